"""Randomised parity sweep THROUGH THE RENDERER on the GPU (not part of the suite: a minute or two): random scenes,
image sizes, K, scalar / 3x3 sigmas -> GaussianRenderer + to_white_background forward and backward (fragments without
act / dsd, fused backward where it applies) against the fp64 oracle chain.  usage: python tests/stress_render.py [n] [seed]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_configs as C
from util import TOL, random_scene

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"image": 0.0, "colors": 0.0, "verts": 0.0, "sigmas": 0.0}
for case in range(n_cases):
    N = int(rng.integers(50, 2500)); H = int(rng.integers(8, 80)); W = int(rng.integers(8, 80))
    K = int(rng.choice([2, 4, 6, 8, 12, 16, 20, 26, 40, 64, 128, 7, 25]))
    aniso = bool(rng.integers(0, 3) == 0)
    verts, sig, cols = random_scene(N, seed=int(rng.integers(1 << 30)), aniso=aniso, lo=0.05, hi=0.2)
    if aniso:
        sig = (0.5 * (sig + sig.transpose(0, 2, 1))).astype(np.float32)
    sc = dict(verts=verts, sigmas=sig, colors=cols, focal=float(rng.uniform(0.7, 1.4)) * max(H, W), principal=(W / 2.0, H / 2.0),
              image_size=(H, W), dist=float(rng.uniform(2.6, 4.0)), elev=float(rng.uniform(-40, 40)), azim=float(rng.uniform(0, 360)), K=K)
    frag, img, gm, colors, (R, T) = C._render(sc)
    ref = C._oracle_frame(sc, R, T)
    same = C._check_frame(f"stress {case}", frag, img, ref, max_flips=max(3, H * W // 300))
    g_img = rng.normal(size=ref["image"].shape) * same[..., None]          # flipped pixels carry no loss
    (img * C.t(g_img)).sum().backward()
    want = C._oracle_grads(sc, ref, g_img)
    got = (colors.grad, gm.verts.grad, gm.sigmas.grad)
    errs = C._check_grads(f"stress {case} N={N} {H}x{W} K={K} {'aniso' if aniso else 'iso'} [{type(img.grad_fn).__name__}]", got, want, 5)
    worst["image"] = max(worst["image"], float(np.abs(C.n(img)[same] - ref["image"][same]).max(initial=0.0)))
    for k, v in errs.items():
        worst[k] = max(worst[k], v)
print("all", n_cases, "cases ok; worst errors / scale:", {k: f"{v:.1e}" for k, v in worst.items()})
