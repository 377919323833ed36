"""Where the sweep's workgroups land (debug build -DVOGE_SWEEP_TIMES): launch index -> (XCD/SE, CU, SIMD), per-SIMD load.
usage on the GPU box: VOGE_HIP_LIB=build/variants/times.so python tools/sweep_hwmap.py"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from voge_amd import _lib, scenes
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from voge_amd.Renderer import GaussianRenderSettings, GaussianRenderer
from voge_amd.Meshes import GaussianMeshes
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
verts, sig, colors = scenes.random_gaussians(N, seed=0)
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
nwg = ((W + 7) // 8) * ((H + 7) // 8)
buf = (ctypes.c_ulonglong * (8 * nwg))()
ctypes.CDLL(_lib.LIB_PATH).voge_debug_sweep_times(buf, nwg)
t = np.array(list(buf), dtype=np.uint64).reshape(nwg, 8)
ran = t[:, 0] > 0
t = t[ran]
blk = (t[:, 6] >> np.uint64(32)).astype(np.int64)
ev = (t[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
hw = t[:, 7].astype(np.int64)
wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
xcc = (t[:, 7].astype(np.uint64) >> np.uint64(32)).astype(np.int64) & 15     # XCC_ID register (gfx94x+)
start = (t[:, 0] - t[:, 0].min()).astype(np.float64) * 0.01
dur = (t[:, 5] - t[:, 0]).astype(np.float64) * 0.01
o = np.argsort(blk)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print("first 24 workgroups by launch index: (blk, xcc, se, sh, cu, simd, wave, start us, dur us, evals)")
for i in o[:24]:
    print(int(blk[i]), int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]), int(wave[i]), round(start[i], 1), round(dur[i], 1), int(ev[i]))
pos = {int(b): i for i, b in enumerate(blk)}
for step in (256, 512, 1024, 2048):
    pairs = [(b, b + step) for b in range(0, 256) if b in pos and b + step in pos]
    same_simd = sum(1 for a, c in pairs if key[pos[a]] == key[pos[c]])
    same_cu = sum(1 for a, c in pairs if key[pos[a]] // 4 == key[pos[c]] // 4)
    print(f"blocks b and b+{step}: same SIMD {same_simd} / {len(pairs)}, same CU {same_cu} / {len(pairs)}")
first = start < 1.0
print("distinct (se, sh, cu, simd) seen:", len(np.unique(key)), " workgroups resident at t < 1 us:", int(first.sum()))
cnt = np.bincount(key[first])
print("waves per SIMD slot among the first wave of residents: histogram", np.bincount(cnt[cnt > 0]))
load = np.zeros(key.max() + 1)
np.add.at(load, key, ev)
l = load[load > 0]
print("evaluations per SIMD slot: mean %.0f  max %.0f  min %.0f  (max / mean %.2f)" % (l.mean(), l.max(), l.min(), l.max() / l.mean()))
