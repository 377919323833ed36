"""What would splitting the sweep's heaviest tiles over two waves buy?  Per-tile durations from the sweep's timers (debug build:
tools/tune_variants.sh bst:"-DVOGE_BIN_TIMES -DVOGE_SWEEP_STATS"), list-scheduled heaviest-first on the 2 304 one-wave slots of today's
kernel; a tile longer than T becomes two jobs of (share x duration + merge) each.  The shares are the model's assumption: a half walks
`share` of the whole tile's depth (0.5 = a perfect shared exit bound; 1.0 = each half walks as deep as the whole, no gain).
usage (GPU box): VOGE_HIP_LIB=build/variants/bst.so [FULL=1] python tools/split_tile_sim.py"""
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
if FULL:
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
nt = ((W + 7) // 8) * ((H + 7) // 8)
buf = (ctypes.c_ulonglong * (8 * nt))()
raw.voge_debug_sweep_times(buf, nt)
ts = np.array(list(buf), dtype=np.float64).reshape(nt, 8) * 0.01
ran = ts[:, 0] > 0
d = (ts[:, 5] - ts[:, 0])[ran]
span = ts[ran, 5].max() - ts[ran, 0].min()


def makespan(jobs, slots=2304):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for x in sorted(jobs, reverse=True):
        t1 = heapq.heappop(free) + x
        end = max(end, t1)
        heapq.heappush(free, t1)
    return end


print(f"{name} ({'entry with act / dsd' if FULL else 'renderer form'}): {len(d)} swept tiles, mean {d.mean():.1f} / max {d.max():.1f} us, measured span {span:.1f} us; "
      f"the same tiles list-scheduled unsplit: {makespan(list(d)):.1f} us; sum / slots {d.sum() / 2304:.1f}")
for share in (0.5, 0.65, 0.8):
    for merge in (1.5, 3.0):
        row = []
        for T in (16.0, 20.0, 24.0):
            jobs = []
            for x in d:
                if x > T:
                    jobs += [share * x + merge] * 2
                else:
                    jobs.append(x)
            row.append(f"T={T:.0f}: {makespan(jobs):.1f}")
        print(f"  a half walks {share:.2f} of the tile, merge {merge:.1f} us:  " + "   ".join(row))
