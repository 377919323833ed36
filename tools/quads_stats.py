"""The sweep's per-tile phases (debug build -DVOGE_SWEEP_TIMES): candidates evaluated per tile (4 x trips), time staging chunks,
time in the trip loops, prologue and epilogue, summed over the lit tiles of the cfg3 frame.
usage on the GPU box: VOGE_HIP_LIB=build/variants/times_q.so python tools/quads_stats.py"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from voge_amd import _lib, scenes
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from voge_amd.Renderer import GaussianRenderSettings, GaussianRenderer
from voge_amd.Meshes import GaussianMeshes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
import os
verts, sig, colors = scenes.random_gaussians(N, seed=0, anisotropic=({'1': True, 'diag': 'diag'}.get(os.environ.get('ANISO', ''), False)))      # (ANISO=1: [N,3,3] forms, ANISO=diag: (N,3))
dev = torch.device("cuda", 0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
_lib.load()
with torch.no_grad():
    for _ in range(4):
        renderer(gm, R=R, T=T)
torch.cuda.synchronize()
nwg = min(8192, ((W + 7) // 8) * ((H + 7) // 8))
buf = (ctypes.c_ulonglong * (8 * nwg))()
ctypes.CDLL(_lib.LIB_PATH).voge_debug_sweep_times(buf, nwg)
t = np.array(list(buf), dtype=np.uint64).reshape(nwg, 8)
t = t[t[:, 0] > 0]
ev = (t[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
f = lambda a: a.astype(np.float64) * 0.01      # (100 MHz wall clock -> us)
pro, fill, cons, epi, tot = f(t[:, 1] - t[:, 0]), f(t[:, 2]), f(t[:, 3]), f(t[:, 5] - t[:, 4]), f(t[:, 5] - t[:, 0])
print(f"{name}: {len(t)} lit tiles; candidates evaluated per tile {ev.mean():.1f} (max {ev.max()}); per tile us: prologue {pro.mean():.2f}  "
      f"staging {fill.mean():.2f}  trips {cons.mean():.2f}  epilogue {epi.mean():.2f}  whole {tot.mean():.2f} (max {tot.max():.1f}); "
      f"launch span {f(t[:, 5].max() - t[:, 0].min()):.1f} us; ns per trip of four {1e3 * cons.sum() / max(ev.sum() / 4, 1):.0f}")
if os.environ.get("SLOW"):      # (-DVOGE_SWEEP_TIMES -DVOGE_SWEEP_SLOW builds: stamp 7 = cycles in the insertions << 32 | their number; stamp 2 = walk statistics)
    ins_t, ins_n = f(t[:, 7] >> np.uint64(32)), (t[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
    lanes = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    moved = ((t[:, 2] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
    far = (t[:, 2] >> np.uint64(48)).astype(np.int64)
    print(f"  insertions (the out-of-line path of commit): {ins_n.mean():.1f} events per tile = {ins_n.sum() / max(ev.sum(), 1):.2f} per candidate, "
          f"{ins_t.mean():.2f} us per tile = {100 * ins_t.sum() / cons.sum():.0f} % of the trip loops, {1e3 * ins_t.sum() / max(ins_n.sum(), 1):.0f} ns per event; "
          f"lanes inserting per event {lanes.sum() / max(ins_n.sum(), 1):.1f}, longest walk per event {moved.sum() / max(ins_n.sum(), 1):.2f} rows, events with a walk of >= 4 rows {100 * far.sum() / max(ins_n.sum(), 1):.0f} %")
