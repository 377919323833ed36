"""The strongly anisotropic scene of tests/test_gpu_parity.py::test_trace_fwd_strongly_anisotropic_ellipsoid_culling (needles and pancakes,
camera inside the cloud) through the current general sweep and through round 3's (the tests' -DVOGE_AB build): trace entry times.
usage (GPU box): python tools/needle_scene_ab.py"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from voge_amd import _lib, ops, scenes
from oracle import camera_np
N, H, W, K = 30000, 384, 416, 24
verts, sig, _ = scenes.random_gaussians(N, seed=5, anisotropic=True, r_lo=0.02, r_hi=0.05, extent=1.2)
rng = np.random.default_rng(9)
nz = rng.normal(size=sig.shape)
sig = (sig + 2e-3 * (nz - nz.swapaxes(-1, -2)) * sig[:, 0:1, 0:1]).astype(np.float32)
sc = dict(verts=verts, sigmas=sig, focal=300.0, principal=(W / 2.0, H / 2.0), image_size=(H, W), dist=1.5, elev=20.0, azim=40.0)
R, T = camera_np.look_at_view_transform([sc["dist"]], [sc["elev"]], [sc["azim"]])
rays, origin = camera_np.pixel_rays(R, T, sc["focal"], sc["principal"], sc["image_size"])
mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
rays = rays.astype(np.float32)
thr_act = oracle.thr_act_of(0.01)
dev = torch.device("cuda", 0)
tm, tA, tr = (torch.tensor(x, device=dev) for x in (mus.reshape(-1, 3), isg.reshape(-1, 3, 3), rays))


def timed(label):
    with torch.no_grad():
        for _ in range(2):
            out = ops.ray_trace_fine(tm, tA, tr, None, thr_act, 16, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = ops.ray_trace_fine(tm, tA, tr, None, thr_act, 16, K)
        e1.record(); torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us per trace entry")
    return [x.cpu().numpy() for x in out]


_lib.load()
a = timed("current general sweep (sweep_iso_kernel<true>)")
ctx = _lib.using(_lib.AB_LIB_PATH)
ab = ctx.__enter__()
ab.voge_debug_sweep_variant.restype, ab.voge_debug_sweep_variant.argtypes = ctypes.c_int, [ctypes.c_int]
try:
    assert ab.voge_debug_sweep_variant(1) == 0
    b = timed("round 3's general sweep (trace_fwd_kernel<1, false>, A/B build)")
finally:
    ab.voge_debug_sweep_variant(0)
    ctx.__exit__()
print("same bits:", all(np.array_equal(x, y) for x, y in zip(a, b)))
