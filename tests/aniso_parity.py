"""Strongly anisotropic Gaussians (random_gaussians(anisotropic=True): needles and pancakes) at the cfg3
size: the full-frame GPU trace against the CPU oracle on a few pixel crops (the oracle is brute force, so
it is cropped to stay in seconds).  Exercises the ellipsoid culling and its depth bound.
usage: python tests/aniso_parity.py [n_crops] [N]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import camera_np
from util import compare_trace
from voge_amd import ops, scenes

n_crops = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N0, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
N = int(sys.argv[2]) if len(sys.argv) > 2 else N0
verts, sig, _ = scenes.random_gaussians(N, seed=0, anisotropic=True)
R, T = camera_np.look_at_view_transform([dd], [el], [az])
rays, origin = camera_np.pixel_rays(R, T, focal, pp, (H, W))
mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
isg = (2 * sig).astype(np.float32)[None]
thr_act = oracle.thr_act_of(0.01)
t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")
sel = ops.ray_trace_fine(t(mus.reshape(-1, 3)), t(isg.reshape(-1, 3, 3)), t(rays), None, thr_act, 10, K)
got = [x.cpu().numpy() for x in sel]
print("hits per pixel: mean", float((got[0] >= 0).sum(-1).mean()))
rng = np.random.default_rng(0)
S = 24
for c in range(n_crops):
    y0, x0 = int(rng.integers(0, H - S)), int(rng.integers(0, W - S))
    ref = oracle.trace_fwd(mus, isg, np.ascontiguousarray(rays[:, y0:y0 + S, x0:x0 + S]), K, thr_act)
    frac = compare_trace([g[:, y0:y0 + S, x0:x0 + S] for g in got], ref, thr_act, min_match=0.97)
    print(f"crop ({y0},{x0}) {S}x{S}: index lists identical on {frac:.4f} of pixels, rest on decision boundaries")
print("ok")
