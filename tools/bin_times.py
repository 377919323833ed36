"""Phase timestamps of the two bin kernels (debug build: tools/tune_variants.sh bt:"-DVOGE_BIN_TIMES").
usage on the GPU box: VOGE_HIP_LIB=build/variants/bt.so python tools/bin_times.py [config]"""
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
verts, sig, colors = scenes.random_gaussians(N, seed=0, anisotropic=bool(__import__("os").environ.get("ANISO")))      # ANISO=1: [N,3,3] sigmas
import os
dd = float(os.environ.get("DIST", dd))      # (DIST=16: the small-object case of tools/cliff_scan.py -- binB's long path)
dev = torch.device("cuda", 0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
with torch.no_grad():
    for _ in range(3):
        renderer(gm, R=R, T=T)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
nst = ((W + 31) // 32) * ((H + 31) // 32)
nreg = ((W + 127) // 128) * ((H + 127) // 128) * 16
for which, n, names in ((0, min(nreg, 1024), ["first loads + region cone", "region test + compaction (round 1)", "child tests + id stores", "(barrier)", "(further rounds)"]),
                        (1, min(nst * 4, 1024), ["gather + keys", "reduce + hist + scan", "scatter (+flag suffix)", "tile filter (all waves)", "slots + spill + sentinel tiles"])):
    buf = (ctypes.c_ulonglong * (8 * n))()
    lib.voge_debug_bin_times(buf, which, n)
    t = np.array(list(buf), dtype=np.float64).reshape(n, 8) * 0.01     # us (100 MHz)
    k = len(names) + 1
    d = np.diff(t[:, :k], axis=1)
    print(["binA", "binB"][which], "workgroups", n, " kernel span", round(t[:, k - 1].max() - t[:, 0].min(), 1), "us;  per workgroup mean / max (us):")
    for i, nm in enumerate(names):
        print(f"  {nm:42s} {d[:, i].mean():6.2f} {d[:, i].max():6.2f}")
    tot = t[:, k - 1] - t[:, 0]
    print(f"  {'total':42s} {tot.mean():6.2f} {tot.max():6.2f}   start spread {t[:, 0].max() - t[:, 0].min():.2f}; first stamp after kernel start")
    if which == 0:
        print(f"  entry -> region cone in LDS (wave 0) {(t[:, 6] - t[:, 0]).mean():6.2f} {(t[:, 6] - t[:, 0]).max():6.2f};  -> first Gaussians loaded, records derived "
              f"{(t[:, 7] - t[:, 6]).mean():6.2f} {(t[:, 7] - t[:, 6]).max():6.2f};  -> through the barrier {(t[:, 1] - t[:, 7]).mean():6.2f} {(t[:, 1] - t[:, 7]).max():6.2f}")
    if which == 1:
        print(f"  entry -> cones done {(t[:, 6] - t[:, 7]).mean():6.2f} {(t[:, 6] - t[:, 7]).max():6.2f};  cones -> rank done {(t[:, 0] - t[:, 6]).mean():6.2f} {(t[:, 0] - t[:, 6]).max():6.2f}"
              f";  first entry -> last end {t[:, k - 1].max() - t[:, 7].min():.1f};  entry spread {t[:, 7].max() - t[:, 7].min():.2f}")
        o = np.argsort(-tot)[:5]
        print("  slowest workgroups:", [(int(i), [round(float(x), 1) for x in d[i]]) for i in o])
