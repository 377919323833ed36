"""Replay ONE case of tests/stress_render.py (seed, case; pattern attr_and_silhouette) and show where the sigma gradient
differs from the fp64 oracle chain: the worst entries, their Gaussians' Sigma^-1 eigenvalues, and the same comparison for
the oracle's own chain fed with the GPU's fp32 forward values.  usage: python tools/stress_grad_case.py <seed> <case>"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import camera_np
import test_gpu_configs as C
from util import random_scene

from stress_render import _well_conditioned
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    N = int(rng.integers(50, 2500)); H = int(rng.integers(8, 80)); W = int(rng.integers(8, 80))
    K = int(rng.choice([2, 4, 6, 8, 12, 16, 20, 26, 40, 64, 128, 7, 25, 1, 33, 130, 200]))
    form = ("scalar", "scalar", "full", "diag")[int(rng.integers(0, 4))]
    pattern = ("white_background", "attr_and_silhouette")[int(rng.integers(0, 2))]
    verts, sig, cols = random_scene(N, seed=int(rng.integers(1 << 30)), aniso=(form == "full"), lo=0.05, hi=0.2)
    if form == "full":
        sig = (0.5 * (sig + sig.transpose(0, 2, 1))).astype(np.float32)
        sig = _well_conditioned(sig)
    elif form == "diag":
        sig = (sig[:, None] * rng.uniform(0.6, 1.6, (N, 3))).astype(np.float32)
    sc = dict(verts=verts, sigmas=sig, colors=cols, focal=float(rng.uniform(0.7, 1.4)) * max(H, W), principal=(W / 2.0, H / 2.0),
              image_size=(H, W), dist=float(rng.uniform(2.6, 4.0)), elev=float(rng.uniform(-40, 40)), azim=float(rng.uniform(0, 360)), K=K)
    views = None
    if rng.random() < 0.35:
        nv = int(rng.integers(2, 4))
        views = camera_np.look_at_view_transform([sc["dist"] + 0.3 * v for v in range(nv)], [sc["elev"] - 15.0 * v for v in range(nv)],
                                                 [sc["azim"] + 70.0 * v for v in range(nv)])
    nB = 1 if views is None else len(views[0])
    if pattern == "white_background":
        g_img = rng.normal(size=(nB, H, W, 3))
    else:
        g_rgb = rng.normal(size=(nB, H, W, 3)); g_silh = rng.normal(size=(nB, H, W))
print(f"case {want}: N={N} {H}x{W} K={K} {form} {pattern} views={nB}")
assert pattern == "attr_and_silhouette" and nB == 1
from voge_amd.Renderer import get_silhouette, interpolate_attr
frag, img, gm, colors, (R, T) = C._render(sc, views=views)
ref = C._oracle_frame(sc, R, T)
same = C._check_frame("case", frag, img, ref, max_flips=10 ** 9)
x_white = ref["rgb"] + (1 - ref["silhouette"])[..., None]
same = same & ~((np.abs(x_white - 1) < 1e-5).any(-1) | (np.abs(ref["weight"].sum(-1) - 1) < 1e-5))
g_rgb = g_rgb * same[..., None]; g_silh = g_silh * same
rgb, sil = interpolate_attr(frag, colors), get_silhouette(frag)
((rgb * C.t(g_rgb)).sum() + (sil * C.t(g_silh)).sum()).backward()
wsum = ref["weight"].sum(-1)
g_attr, g_w = oracle.merge_bwd(ref["colsB"], ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
live = np.arange(K)[None, None, None] < ref["valid_num"][..., None]
g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w + (g_silh * (wsum < 1))[..., None] * live, 1.0)
_, g_mu, g_A = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
want_sig = 2 * g_A.reshape(N, 3, 3)
got = C.n(gm.sigmas.grad).astype(np.float64)
err = np.abs(got - want_sig)
scale = np.abs(want_sig).max()
print(f"sigma gradient: max |err| {err.max():.3e}, scale {scale:.3e}, ratio {err.max() / scale:.2e}")
per = err.reshape(N, 9).max(1)
for g in np.argsort(-per)[:5]:
    ev = np.linalg.eigvalsh(sig[g].astype(np.float64))
    hits = int((ref["idx"] == g).sum())
    print(f"  Gaussian {g}: err {per[g]:.3e}, |grad| {np.abs(want_sig[g]).max():.3e}, Sigma^-1 eigenvalues {ev}, pixels hit {hits}, "
          f"|mu| {np.linalg.norm(np.asarray(ref['mus']).reshape(-1, 3)[g]):.2f}, worst entry got {got[g].reshape(-1)[np.argmax(err[g])]:.5e} want {want_sig[g].reshape(-1)[np.argmax(err[g])]:.5e}")
