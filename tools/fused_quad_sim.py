"""What would a fused binB + sweep kernel (one workgroup per 16x16-px quad -- bin phase, then its four tiles swept
by its four waves) take?  Measured per-workgroup durations of today's two kernels on the same frame (debug build with both timers:
tools/tune_variants.sh bst:"-DVOGE_BIN_TIMES -DVOGE_SWEEP_STATS"), then a list-scheduling simulation of the fused launch: a quad's
workgroup lasts (its binB workgroup's duration) + (its slowest tile's duration) and holds one of `slots` residency slots -- the
sweep's top-K lists are 17.7 KB of LDS per tile, 71 KB per quad: two quads per CU, 512 slots.  Optimistic: nothing is charged for
sharing a CU, for the lost heaviest-first tile order inside a quad, or for finding the order.
usage (GPU box): VOGE_HIP_LIB=build/variants/bst.so [FULL=1] python tools/fused_quad_sim.py"""
import ctypes, heapq, math, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from voge_amd import _lib, scenes, ops
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from voge_amd.Renderer import GaussianRenderSettings, GaussianRenderer
from voge_amd.Meshes import GaussianMeshes

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, colors = scenes.random_gaussians(N, seed=0)
dev = torch.device("cuda", 0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
FULL = bool(os.environ.get("FULL"))
if FULL:      # the stand-alone entry (idx, len, act, dsd)
    from oracle import camera_np
    Rn, Tn = camera_np.look_at_view_transform([dd], [el], [az])
    rays_np, origin = camera_np.pixel_rays(Rn, Tn, focal, pp, (H, W))
    mus_t = torch.from_numpy((verts - origin[0].astype("float32")).astype("float32")).to(dev)
    a_t = torch.from_numpy((2 * sig).astype("float32")).to(dev)
    rays_t = torch.from_numpy(rays_np.astype("float32")).to(dev)
    run = lambda: ops._RayTraceVoGEIso.apply(mus_t, a_t, rays_t, None, -math.log(0.01 + 1e-10), K)
else:
    run = lambda: renderer(gm, R=R, T=T)
with torch.no_grad():
    for _ in range(4):
        run()
torch.cuda.synchronize()
tick = 0.01
nstx, nsty = (W + 31) // 32, (H + 31) // 32
nq = nstx * nsty * 4
assert nq <= 1024, "the bin timers keep 1024 workgroups"
buf = (ctypes.c_ulonglong * (8 * nq))()
raw.voge_debug_bin_times(buf, 1, nq)
tb = np.array(list(buf), dtype=np.float64).reshape(nq, 8) * tick
d_bin = tb[:, 5] - tb[:, 7]                       # kernel entry (stamp 7) -> last phase done (stamp 5)
span_bin = tb[:, 5].max() - tb[:, 7].min()
ntx, nty = (W + 7) // 8, (H + 7) // 8
nt = ntx * nty
buf = (ctypes.c_ulonglong * (8 * nt))()
raw.voge_debug_sweep_times(buf, nt)
ts = np.array(list(buf), dtype=np.float64).reshape(nt, 8) * tick
ran = ts[:, 0] > 0
d_tile = np.where(ran, ts[:, 5] - ts[:, 0], 0.0)
span_sw = ts[ran, 5].max() - ts[ran, 0].min()
d_sw = np.zeros(nq)
for ty in range(nty):
    for tx in range(ntx):
        q = ((ty // 4) * nstx + tx // 4) * 4 + ((ty % 4) // 2) * 2 + (tx % 4) // 2
        d_sw[q] = max(d_sw[q], d_tile[ty * ntx + tx])
D = d_bin + d_sw
print(f"{name} ({'entry with act / dsd' if FULL else 'renderer form'}): binB workgroups {d_bin.mean():.1f} us mean / {d_bin.max():.1f} max, kernel span {span_bin:.1f}; "
      f"swept tiles {int(ran.sum())}, {d_tile[ran].mean():.1f} mean / {d_tile.max():.1f} max, kernel span {span_sw:.1f}; today {span_bin + span_sw:.1f} us for the two")
print(f"fused workgroup = bin phase + its slowest tile: {D.mean():.1f} us mean, {D.max():.1f} max; sum / slots = {D.sum() / 512:.1f} us at 512 slots")


def makespan(order, slots):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for q in order:
        t0 = heapq.heappop(free)
        t1 = t0 + D[q]
        end = max(end, t1)
        heapq.heappush(free, t1)
    return end


for slots in (512, 768):
    print(f"  {slots} slots ({slots // 256} quads per CU): quads in image order {makespan(range(nq), slots):.1f} us;  heaviest first (an order nobody has "
          f"before binB has run) {makespan(np.argsort(-D), slots):.1f} us;  by the bin phase's own duration first {makespan(np.argsort(-d_bin), slots):.1f} us")
