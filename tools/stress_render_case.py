"""Replay ONE case of tests/stress_render.py (seed, case) with both scalar-sigma sweep kernels and count, for each, the
pixels whose index list differs from the fp64 oracle's.  usage: python tools/stress_render_case.py <seed> <case>"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import camera_np
import test_gpu_configs as C
from util import random_scene
from voge_amd import _lib

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
    if case < want:      # (the cases in front consume random numbers inside their bodies too: replay them)
        if pattern == "white_background":
            rng.normal(size=((1 if views is None else len(views[0])), H, W, 3))
        else:
            nB = 1 if views is None else len(views[0])
            rng.normal(size=(nB, H, W, 3)); rng.normal(size=(nB, H, W))
print(f"case {want}: N={N} {H}x{W} K={K} {form} {pattern} views={None if views is None else len(views[0])}")
import ctypes
_ctx = _lib.using(_lib.AB_LIB_PATH)      # (the -DVOGE_AB build: both sweeps and the switch between them)
lib = _ctx.__enter__()
lib.voge_debug_sweep_variant.restype, lib.voge_debug_sweep_variant.argtypes = ctypes.c_int, [ctypes.c_int]
out = {}
for v in (0, 1):
    lib.voge_debug_sweep_variant(v)
    frag, img, gm, colors, (R, T) = C._render(sc, views=views)
    ref = C._oracle_frame(sc, R, T)
    idx = C.n(frag.vert_index)
    same = (idx == np.where(ref["idx"] < 0, 0, ref["idx"])).all(-1) | (idx == ref["idx"]).all(-1)
    out[v] = idx
    print(f"variant {v}: {(~same).sum()} of {same.size} pixels differ from the oracle")
    for b, y, x in zip(*np.nonzero(~same)):
        a, r = idx[b, y, x], ref["idx"][b, y, x]
        k = int(np.argmax(a != np.where(r < 0, 0, r)))
        thr = ref["thr_act"]
        print(f"   pixel {(b, y, x)}: first difference at slot {k}: got {a[k]} want {r[k]}; oracle len there {ref['len'][b, y, x, k]:.7f} act {ref['act'][b, y, x, k]:.6f} (thr {thr:.6f}), "
              f"next len {ref['len'][b, y, x, min(k + 1, K - 1)]:.7f}, hits {ref['valid_num'][b, y, x]}")
lib.voge_debug_sweep_variant(0)
print("the two kernels agree:", bool((out[0] == out[1]).all()))
