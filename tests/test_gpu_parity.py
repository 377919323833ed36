"""GPU parity tests: the HIP path (through the C ABI / voge_amd.ops) against the CPU oracle on
the same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's
full sizes -- through size-independent properties.  Tolerance: 1e-4 (north_star), relative to
max(1, |ref|); index lists exact (see util.compare_trace for the boundary rule)."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import camera_np
from util import GOLDEN, TOL, bunny_scene, close, compare_trace, cuboid_scene, grad_close, max_rel, random_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a, dtype=torch.float32, rg=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=rg)


def n(x):
    return x.detach().cpu().numpy()


def camera_inputs(scene, image_size=None, B=1):
    size = image_size or scene["image_size"]
    azims = [scene["azim"] + 25.0 * b for b in range(B)]
    R, T = camera_np.look_at_view_transform([scene["dist"]] * B, [scene["elev"]] * B, azims)
    rays, origin = camera_np.pixel_rays(R, T, scene["focal"], scene["principal"], size)
    verts = np.asarray(scene["verts"], np.float32)
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(np.asarray(scene["sigmas"], np.float32))).astype(np.float32)
    isg = np.ascontiguousarray(np.broadcast_to(isg[None], (B,) + isg.shape))
    return mus, isg, rays, R, T


def run_trace(mus, isg, rays, K, thr_act, bins=None, bin_size=0):
    from voge_amd import ops
    out = ops.ray_trace_fine(t(mus).reshape(-1, 3), t(isg).reshape(-1, 3, 3), t(rays), bins, thr_act, bin_size, K)
    torch.cuda.synchronize()
    return [n(o) for o in out]


# ------------------------------------------------------------------------------- trace fwd
def test_abi_loaded_is_in_tree(hip_lib):
    from voge_amd import _lib
    assert os.path.samefile(os.path.dirname(_lib.LIB_PATH), os.path.join(os.path.dirname(GOLDEN), "..", "voge_amd"))
    assert hip_lib.voge_abi_version() == 7


def test_trace_fwd_cuboid_config1(hip_lib):
    sc = cuboid_scene()
    mus, isg, rays, _, _ = camera_inputs(sc)
    thr = oracle.thr_act_of(0.01)
    got = run_trace(mus, isg, rays, sc["K"], thr)
    ref = oracle.trace_fwd(mus, isg, rays, sc["K"], thr)
    frac = compare_trace(got, ref, thr, max_flips=10)
    assert (ref[0] >= 0).sum() > 100000 and frac > 0.999


def test_trace_fwd_bunny_config2(hip_lib):
    """Ill-conditioned for the reference's own fp32 formula (A ~ 2e4..1e6, distance 6): the
    reference-order fp32 oracle is shown to miss the tolerance while the HIP path meets it."""
    sc = bunny_scene()
    mus, isg, rays, _, _ = camera_inputs(sc)
    thr = oracle.thr_act_of(0.01)
    got = run_trace(mus, isg, rays, sc["K"], thr)
    ref = oracle.trace_fwd(mus, isg, rays, sc["K"], thr)
    compare_trace(got, ref, thr, min_match=0.995, max_flips=100)
    ref32 = oracle.trace_fwd(mus, isg, rays, sc["K"], thr, precision="f32")
    both = (ref32[0] == ref[0]) & (ref[0] >= 0)
    assert max_rel(ref32[2][both], ref[2][both]) > 10 * TOL  # the reference's fp32 noise floor


@pytest.mark.parametrize("H,W,K,B,thr", [(37, 53, 102, 1, 0.0), (16, 16, 5, 2, 0.01), (70, 45, 33, 2, 0.01),
                                         (8, 130, 200, 1, 0.01), (1, 1, 1, 1, 0.01)])
def test_trace_fwd_anisotropic_ragged(hip_lib, H, W, K, B, thr):
    """Full 3x3 Sigma^-1 (EfficientCuboid-style L L^T), non-square / ragged sizes, K up to 200
    (K > number of hits -> sentinel tails), thr_activation=0, batch of 2 views."""
    verts, sig, _ = random_scene(700, seed=H * 1000 + W, aniso=True, lo=0.05, hi=0.15)
    sc = dict(verts=verts, sigmas=sig, focal=0.9 * max(H, W), principal=(W / 2.0, H / 2.0), image_size=(H, W),
              dist=3.5, elev=20.0, azim=-40.0)
    mus, isg, rays, _, _ = camera_inputs(sc, B=B)
    thr_act = oracle.thr_act_of(thr)
    got = run_trace(mus, isg, rays, K, thr_act)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    compare_trace(got, ref, thr_act, min_match=0.99)


@pytest.mark.parametrize("iso_api", [False, True])
def test_trace_fwd_large_frame_coarse_bins_and_tile_order(hip_lib, iso_api):
    """A frame big enough to take the paths small tests skip: the coarse 128x128 binning level
    (super-tiles x Gaussians >= 2^21), more than 2048 sweep tiles (longest-first launch order,
    unbalanced: most of the frame is empty), a batch of two views; mixed isotropic / anisotropic
    Gaussians through the general entry point, all-isotropic through the scalar one."""
    from voge_amd import ops
    N, H, W, K, B = 13000, 392, 408, 10, 2
    verts, sig, _ = random_scene(N, seed=77, lo=0.02, hi=0.05, extent=0.8)
    sc = dict(verts=verts, sigmas=sig, focal=430.0, principal=(W / 2.0, H / 2.0), image_size=(H, W),
              dist=4.5, elev=15.0, azim=-25.0)
    mus, isg, rays, _, _ = camera_inputs(sc, B=B)
    if not iso_api:   # every third Gaussian gets a full symmetric 3x3 form (plus a 1e-3 skew part)
        rng = np.random.default_rng(3)
        isg = isg.copy()
        nz = rng.normal(size=isg[:, ::3].shape)
        nz = 0.1 * (nz + nz.swapaxes(-1, -2)) / 2 + 1e-3 * (nz - nz.swapaxes(-1, -2))
        isg[:, ::3] += (nz * isg[:, ::3, 0:1, 0:1]).astype(np.float32)
    thr_act = oracle.thr_act_of(0.01)
    if iso_api:
        a = np.ascontiguousarray(isg[..., 0, 0])
        got = [n(x) for x in ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)]
    else:
        got = run_trace(mus, isg, rays, K, thr_act)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    compare_trace(got, ref, thr_act, min_match=0.9999, max_flips=30)
    assert (ref[0][..., -1] >= 0).mean() > 0.05 and (ref[0][..., 0] < 0).mean() > 0.3   # full lists and empty pixels


def test_trace_fwd_strongly_anisotropic_ellipsoid_culling(hip_lib):
    """Needles and pancakes (A = L L^T with a random lower-triangular L, EfficientCuboidViaOptimization.py:17-18,
    plus a small skew part) on a frame large enough for all three binning levels: the bin kernels
    cull these by their hit ellipsoid (separating plane against the tile's ray cone) and bound their
    depth by the ellipsoid's extent along the tile axis, which must stay conservative.  The brute-force
    oracle is cropped to a few windows to stay in seconds; behind-the-camera Gaussians are kept (-1 path)."""
    from voge_amd import scenes
    N, H, W, K = 30000, 384, 416, 24
    verts, sig, _ = scenes.random_gaussians(N, seed=5, anisotropic=True, r_lo=0.02, r_hi=0.05, extent=1.2)
    rng = np.random.default_rng(9)
    nz = rng.normal(size=sig.shape)
    sig = (sig + 2e-3 * (nz - nz.swapaxes(-1, -2)) * sig[:, 0:1, 0:1]).astype(np.float32)
    sc = dict(verts=verts, sigmas=sig, focal=300.0, principal=(W / 2.0, H / 2.0), image_size=(H, W),
              dist=1.5, elev=20.0, azim=40.0)   # camera inside the cloud: Gaussians on both sides of it
    mus, isg, rays, _, _ = camera_inputs(sc)
    thr_act = oracle.thr_act_of(0.01)
    got = run_trace(mus, isg, rays, K, thr_act)
    S = 20
    n_neg = 0
    for y0, x0 in ((0, 0), (180, 200), (H - S, W - S), (90, 300), (300, 40)):
        ref = oracle.trace_fwd(mus, isg, np.ascontiguousarray(rays[:, y0:y0 + S, x0:x0 + S]), K, thr_act)
        compare_trace([g[:, y0:y0 + S, x0:x0 + S] for g in got], ref, thr_act, min_match=0.99, max_flips=3)
        n_neg += int(((ref[1] < 0) & (ref[0] >= 0)).sum())
        assert (ref[0][..., -1] >= 0).mean() > 0.5          # lists are full: the early exit is exercised
    assert n_neg > 0                                         # hits behind the camera were part of it


def test_trace_fwd_nonsymmetric_and_behind_camera(hip_lib):
    """All 9 entries of isigmas are independent inputs (ray_trace_voge.cu:11-38), and Gaussians
    behind the camera are kept with negative len in the -1 path (no sign test, :197)."""
    rng = np.random.default_rng(11)
    N, H, W, K = 300, 24, 24, 12
    verts = rng.uniform(-1, 1, (N, 3)).astype(np.float32) * np.float32([1, 1, 6])
    L = np.tril(rng.uniform(0.5, 1.5, (N, 3, 3))) * 6
    sig = (L @ L.transpose(0, 2, 1) + rng.normal(size=(N, 3, 3)) * 0.5).astype(np.float32)
    sc = dict(verts=verts, sigmas=sig, focal=30.0, principal=(12.0, 12.0), image_size=(H, W), dist=1.0, elev=0.0, azim=0.0)
    mus, isg, rays, R, _ = camera_inputs(sc)
    thr_act = oracle.thr_act_of(0.01)
    got = run_trace(mus, isg, rays, K, thr_act)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    compare_trace(got, ref, thr_act, min_match=0.99, max_flips=4)
    assert (ref[1][ref[0] >= 0] < 0).any(), "scene must contain hits behind the camera"
    # coarse-stage candidate rule (rasterize_coarse.cu:35): view-space z < 0 skipped
    from voge_amd import ops
    fwd = t(R[:, :, 2])
    got2 = [n(o) for o in ops.ray_trace_fine(t(mus).reshape(-1, 3), t(isg).reshape(-1, 3, 3), t(rays), fwd, thr_act, 10, K)]
    front = np.nonzero((mus[0] @ R[0][:, 2]) >= 0)[0].astype(np.int32)
    bins = np.broadcast_to(front[None, None, None, :], (1, 1, 1, front.size))
    ref2 = oracle.trace_fwd(mus, isg, rays, K, thr_act, bin_points=bins, bin_size=max(H, W))
    compare_trace(got2, ref2, thr_act, min_match=0.99, max_flips=4)


def test_trace_fwd_explicit_bin_lists(hip_lib):
    """The reference boundary's bin_points argument: [B,BH,BW,M] int32 with -1 padding."""
    rng = np.random.default_rng(2)
    verts, sig, _ = random_scene(400, seed=9, lo=0.05, hi=0.12)
    H, W, K, bs = 45, 61, 9, 10
    sc = dict(verts=verts, sigmas=sig, focal=50.0, principal=(30.0, 22.0), image_size=(H, W), dist=3.0, elev=5.0, azim=15.0)
    mus, isg, rays, _, _ = camera_inputs(sc, B=2)
    BH, BW, M = (H - 1) // bs + 1, (W - 1) // bs + 1, 150
    bins = np.full((2, BH, BW, M), -1, np.int32)
    for b in range(2):
        for y in range(BH):
            for x in range(BW):
                m = rng.integers(0, M)
                pick = rng.choice(400, size=m, replace=False) + 400 * b
                bins[b, y, x, rng.choice(M, size=m, replace=False)] = pick   # unordered, holes
    thr_act = oracle.thr_act_of(0.01)
    got = run_trace(mus, isg, rays, K, thr_act, bins=t(bins, torch.int32), bin_size=bs)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act, bin_points=bins, bin_size=bs)
    compare_trace(got, ref, thr_act, min_match=0.998, max_flips=10)


def test_explicit_bin_lists_tie_break_deviation(hip_lib):
    """The one documented ordering deviation, measured (include/voge_hip.h, voge_trace_topk_list_fwd): EXACT ties in len
    inside an explicit candidate list are kept in ascending index order here, in list order by the reference
    (ray_trace_voge.cu:197-212 inserts with a strict `<` while walking the list).  Exact ties need bit-identical
    Gaussians, so the scene is 60 Gaussians listed twice (indices i and i + 60) in DESCENDING index order: the oracle
    (reference rule) then puts i + 60 in front of i, the kernel i in front of i + 60.  Everything else -- which
    Gaussians are kept up to the twin, every len / act / dsd value, and the composite built from them -- is identical."""
    n0 = 60
    verts, sig, _ = random_scene(n0, seed=21, lo=0.08, hi=0.15)
    verts, sig = np.concatenate([verts, verts]), np.concatenate([sig, sig])
    H, W, K, bs = 32, 40, 12, 10
    sc = dict(verts=verts, sigmas=sig, focal=40.0, principal=(20.0, 16.0), image_size=(H, W), dist=3.0, elev=5.0, azim=15.0)
    mus, isg, rays, _, _ = camera_inputs(sc, B=1)
    BH, BW = (H - 1) // bs + 1, (W - 1) // bs + 1
    bins = np.broadcast_to(np.arange(2 * n0 - 1, -1, -1, dtype=np.int32)[None, None, None], (1, BH, BW, 2 * n0)).copy()
    thr_act = oracle.thr_act_of(0.01)
    got = run_trace(mus, isg, rays, K, thr_act, bins=t(bins, torch.int32), bin_size=bs)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act, bin_points=bins, bin_size=bs)
    gi, ri = np.asarray(got[0]), np.asarray(ref[0])
    live = ri >= 0
    assert ((gi >= 0) == live).all()
    assert (gi[live] % n0 == ri[live] % n0).all()                       # the same Gaussian up to its twin, slot by slot
    for g, r in zip(got[1:], ref[1:]):                                   # and the same values in every slot
        assert close(np.asarray(g)[live], np.asarray(r)[live]).all()
    differs = (gi != ri).any(-1)
    print(f"[parity] explicit lists, exact ties: {int(differs.sum())} of {differs.size} pixels order a tied pair differently")
    assert differs.any()                                                 # the deviation is real ...
    g2, r2, lv = gi.reshape(-1, K), ri.reshape(-1, K), live.reshape(-1, K)
    both = lv[:, :-1] & lv[:, 1:]
    tw_g = both & (g2[:, :-1] % n0 == g2[:, 1:] % n0)                    # slots k and k + 1 hold twins
    tw_r = both & (r2[:, :-1] % n0 == r2[:, 1:] % n0)
    assert tw_g.any() and (g2[:, :-1][tw_g] < g2[:, 1:][tw_g]).all()     # ... ascending index here
    assert tw_r.any() and (r2[:, :-1][tw_r] > r2[:, 1:][tw_r]).all()     # list order (descending indices) there
    # twins are interchangeable downstream: the composite weights agree slot by slot
    from voge_amd import ops
    w, _ = ops.composite(t(gi, torch.int32), t(got[2]), t(got[1]), t(got[3]), 1.0)
    wr, _ = oracle.composite_fwd(ri, ref[2], ref[1], ref[3], 1.0)
    assert np.abs(n(w) - wr).max() < TOL


def test_trace_fwd_empty_inputs(hip_lib):
    from voge_amd import ops
    rays = t(camera_np.pixel_rays(*camera_np.look_at_view_transform(3, 0, 0), 20.0, (4.0, 4.0), (8, 8))[0])
    idx, ln, act, dsd = ops.ray_trace_fine(torch.zeros((0, 3), device=DEV), torch.zeros((0, 3, 3), device=DEV), rays,
                                           None, 4.6, 10, 7)
    assert (idx == -1).all() and (ln == 1e10).all() and (act == 1e10).all() and (dsd == 0).all()
    w, vn = ops.composite(idx, act, ln, dsd, 1.0)
    assert (w == 0).all() and (vn == 0).all()
    with pytest.raises(RuntimeError):
        ops.ray_trace_fine(torch.zeros((4, 3)), torch.zeros((4, 3, 3)), rays.cpu(), None, 4.6, 10, 7)  # CPU tensors


# ------------------------------------------------------------------------------- trace bwd
def test_trace_bwd_known_answer(hip_lib):
    from voge_amd import ops
    k = np.load(os.path.join(GOLDEN, "trace_bwd_known_answer.npz"))
    mu, A, ray = t(k["mu"][None], rg=True), t(k["isigma"][None], rg=True), t(k["ray"].reshape(1, 1, 1, 3), rg=True)
    idx, ln, act, dsd = ops.ray_trace_fine(mu, A, ray, None, 1e9, 10, 1)
    (ln + act).sum().backward()
    assert np.abs(n(mu.grad)[0] - [1.0, 0.78, 0.0]).max() < 1e-5
    assert np.abs(n(A.grad)[0] - [[-0.04, 0.24, 0], [-0.18, 0.04, 0], [0, 0, 0]]).max() < 1e-5
    assert np.abs(n(ray.grad).reshape(3) - [-1.3, -0.944, 0.0]).max() < 1e-5


@pytest.mark.parametrize("aniso", [False, True])
def test_trace_bwd_vs_oracle(hip_lib, aniso):
    from voge_amd import ops
    verts, sig, _ = random_scene(500, seed=21, aniso=aniso, lo=0.06, hi=0.15)
    if aniso:
        sig = sig + np.random.default_rng(1).normal(size=sig.shape).astype(np.float32) * 0.02 * np.abs(sig).mean()
    H, W, K = 40, 56, 14
    sc = dict(verts=verts, sigmas=sig, focal=45.0, principal=(28.0, 20.0), image_size=(H, W), dist=3.2, elev=-10.0, azim=30.0)
    mus, isg, rays, _, _ = camera_inputs(sc)
    thr_act = oracle.thr_act_of(0.01)
    tm, tA, tr = t(mus.reshape(-1, 3), rg=True), t(isg.reshape(-1, 3, 3), rg=True), t(rays, rg=True)
    idx, ln, act, dsd = ops.ray_trace_fine(tm, tA, tr, None, thr_act, 10, K)
    rng = np.random.default_rng(4)
    gl, ga, gd = (rng.normal(size=idx.shape) for _ in range(3))
    valid = n(idx) >= 0
    gl, ga, gd = gl * valid, ga * valid, gd * valid * 1e-3
    (ln * t(gl) + act * t(ga) + dsd * t(gd)).sum().backward()
    g_ray, g_mu, g_A = oracle.trace_bwd(mus, isg, rays, n(idx), gl, ga, gd)
    # float atomics accumulate ~100s of terms: compare relative to the gradient scale
    for name, got, ref in (("mus", tm.grad, g_mu), ("isg", tA.grad, g_A), ("rays", tr.grad, g_ray)):
        err = np.abs(n(got).astype(np.float64) - ref).max()
        assert err <= TOL * max(1.0, np.abs(ref).max()), f"{name}: {err:.3e} vs scale {np.abs(ref).max():.3e}"


def test_trace_iso_scalar_form_vs_oracle(hip_lib):
    """Isotropic Gaussians kept as one scalar each (the (N,) sigma form): forward equals the
    general trace on a*I, the backward returns d/da = trace(g_A) and the same g_mus / g_ray."""
    from voge_amd import ops
    verts, sig, _ = random_scene(600, seed=33, lo=0.06, hi=0.15)            # sig: (N,) scalars
    assert sig.ndim == 1
    H, W, K, B = 40, 56, 14, 2
    sc = dict(verts=verts, sigmas=sig, focal=45.0, principal=(28.0, 20.0), image_size=(H, W), dist=3.2, elev=-10.0, azim=30.0)
    mus, isg, rays, _, _ = camera_inputs(sc, B=B)                             # isg = 2 * sig * I, [B,N,3,3]
    a = np.ascontiguousarray(isg[..., 0, 0])                                  # [B,N]
    thr_act = oracle.thr_act_of(0.01)
    tm, ta, tr = t(mus.reshape(-1, 3), rg=True), t(a.reshape(-1), rg=True), t(rays, rg=True)
    idx, ln, act, dsd = ops._RayTraceVoGEIso.apply(tm, ta, tr, None, thr_act, K)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    compare_trace([n(x) for x in (idx, ln, act, dsd)], ref, thr_act, min_match=0.998, max_flips=8)
    rng = np.random.default_rng(5)
    valid = n(idx) >= 0
    gl, ga, gd = (rng.normal(size=idx.shape) * valid for _ in range(3))
    gd = gd * 1e-3
    (ln * t(gl) + act * t(ga) + dsd * t(gd)).sum().backward()
    g_ray, g_mu, g_A = oracle.trace_bwd(mus, isg, rays, n(idx), gl, ga, gd)
    g_a = np.einsum("nii->n", g_A.reshape(-1, 3, 3))
    for name, got, want in (("mus", tm.grad, g_mu), ("a", ta.grad, g_a), ("rays", tr.grad, g_ray)):
        err = np.abs(n(got).astype(np.float64).reshape(want.shape) - want).max()
        assert err <= TOL * max(1.0, np.abs(want).max()), f"{name}: {err:.3e} vs scale {np.abs(want).max():.3e}"


def test_trace_iso_records_in_their_own_pass(hip_lib):
    """From 131 072 Gaussians per batch element on, the scalar-sigma entry points make the (centre, reach) / (centre, a) records in
    a pass of their own and binA reads them (trace_fwd.hip: iso_prep_kernel) instead of deriving them per region.  Two views with
    their own 140 000 Gaussians each (the pass's batch axis), a fused preamble (origin subtraction, a = 2 sigma) and the plain
    entry, against the oracle and against each other."""
    from voge_amd import ops
    rng = np.random.default_rng(140)
    B, N, H, W, K = 2, 140000, 16, 24, 12
    verts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    sig = rng.uniform(900, 3000, (B, N)).astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.4] * B, [8.0, -15.0], [20.0, 130.0])
    rays, origin = camera_np.pixel_rays(R, T, 40.0, (W / 2.0, H / 2.0), (H, W))
    rays, origin = rays.astype(np.float32), origin.astype(np.float32)
    thr_act = oracle.thr_act_of(0.01)
    out_view = ops._RayTraceVoGEIsoView.apply(t(verts), t(sig), t(origin), t(rays), None, thr_act, K, 1)
    mus = (verts - origin[:, None]).astype(np.float32)
    a = (2.0 * sig).astype(np.float32)
    out_plain = ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)
    for x, y in zip(out_view, out_plain):
        assert torch.equal(x, y)
    isg = a[..., None, None] * np.eye(3, dtype=np.float32)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    assert (ref[0] >= 0).mean() > 0.5
    compare_trace([n(x) for x in out_plain], ref, thr_act, min_match=0.99, max_flips=8)


@pytest.mark.parametrize("general", [False, True])
def test_small_set_path_equals_the_binned_path(hip_lib, general):
    """Up to 4 096 Gaussians per batch element the trace skips binA (binB reads the per-Gaussian records, trace_fwd.hip:
    small_set_marks); one Gaussian more and the same scene goes through binA's segments.  The extra Gaussian sits far behind
    the camera and is never hit: both paths must produce the same bits."""
    from voge_amd import ops
    rng = np.random.default_rng(4096)
    B, N, H, W, K = 2, 4096, 48, 64, 16
    verts = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    sig = rng.uniform(150, 700, N).astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.3] * B, [10.0, -20.0], [30.0, 100.0])
    rays, origin = camera_np.pixel_rays(R, T, 70.0, (W / 2.0, H / 2.0), (H, W))
    rays, origin = rays.astype(np.float32), origin.astype(np.float32)
    thr_act = oracle.thr_act_of(0.01)
    outs = []
    for extra in (0, 1):
        v = np.concatenate((verts, np.full((extra, 3), 40.0, np.float32)))            # (40, 40, 40): nowhere near a ray
        s_ = np.concatenate((sig, np.full(extra, 300.0, np.float32)))
        mus = (v[None] - origin[:, None]).astype(np.float32)                           # [B, N + extra, 3]
        n_all = N + extra
        if general:
            isg = np.broadcast_to((2.0 * s_)[None, :, None, None] * np.eye(3, dtype=np.float32), (B, n_all, 3, 3)).copy()
            isg[:, ::3, 0, 0] *= 1.5                                                    # a third of them anisotropic
            o = ops.ray_trace_fine(t(mus.reshape(-1, 3)), t(isg.reshape(-1, 3, 3)), t(rays), None, thr_act, 16, K)
        else:
            a = np.broadcast_to((2.0 * s_)[None], (B, n_all)).copy()
            o = ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)
        idx = n(o[0]).copy()
        b_of = np.arange(B)[:, None, None, None]
        local = np.where(idx >= 0, idx - b_of * n_all, -1)                             # indices are b * N + n: compare per view
        assert local.max() < N
        outs.append((local, [n(x) for x in o[1:]]))
    assert (outs[0][0] >= 0).mean() > 0.3
    assert np.array_equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(x, y)


# ------------------------------------------------------------------------------- composite
@pytest.mark.parametrize("name", ["k5", "k25", "k40"])
def test_composite_golden(hip_lib, name):
    from voge_amd import ops
    g = np.load(os.path.join(GOLDEN, f"composite_{name}.npz"))
    act, ln, dsd = t(g["act"], rg=True), t(g["len"], rg=True), t(g["dsd"], rg=True)
    w, vn = ops.composite(t(g["idx"], torch.int32), act, ln, dsd, float(g["occ"]))
    assert vn.dtype == torch.int64 and (n(vn) == g["valid_num"]).all()
    assert np.abs(n(w) - g["weight"]).max() < TOL
    assert (n(w)[g["idx"] < 0] == 0).all()
    (w * t(g["g_weight"])).sum().backward()
    for got, key in ((act.grad, "g_act"), (ln.grad, "g_len"), (dsd.grad, "g_dsd")):
        ref = g[key]
        assert np.abs(n(got) - ref).max() <= TOL * max(1.0, np.abs(ref).max()), key


@pytest.mark.parametrize("K", [1, 20, 64, 102, 250, 256])
def test_composite_random_vs_oracle(hip_lib, K):
    from voge_amd import ops
    rng = np.random.default_rng(K)
    npix = 3 * 67
    ln = np.sort(rng.uniform(1, 8, (npix, K)), axis=1)
    act = rng.uniform(0, 6, (npix, K))
    dsd = rng.uniform(1, 3000, (npix, K))
    idx = rng.integers(0, 1000, (npix, K)).astype(np.int32)
    nv = rng.integers(0, K + 1, npix)
    hole = np.arange(K)[None] >= nv[:, None]
    idx[hole], ln[hole], act[hole], dsd[hole] = -1, 1e10, 1e10, 0
    shape = (3, 67, K)
    ta, tl, td = (t(x.reshape(shape), rg=True) for x in (act, ln, dsd))
    w, vn = ops.composite(t(idx.reshape(shape), torch.int32), ta, tl, td, 0.7)
    wr, vr = oracle.composite_fwd(idx, act, ln, dsd, 0.7)
    assert (n(vn).reshape(-1) == vr).all() and np.abs(n(w).reshape(npix, K) - wr).max() < TOL
    gw = rng.normal(size=(npix, K))
    (w * t(gw.reshape(shape))).sum().backward()
    ra, rl, rd = oracle.composite_bwd(act.astype(np.float32), ln.astype(np.float32), dsd.astype(np.float32), gw, 0.7)
    for got, ref in ((ta.grad, ra), (tl.grad, rl), (td.grad, rd)):
        assert np.abs(n(got).reshape(npix, K) - ref).max() <= TOL * max(1.0, np.abs(ref).max())
    # the same backward through the C ABI without the forward's weights (recompute mode)
    from voge_amd import _lib
    lib = _lib.load()
    tg = t(gw.reshape(shape))
    outs = [torch.empty_like(ta) for _ in range(3)]
    rc = lib.voge_composite_bwd(ta.data_ptr(), tl.data_ptr(), td.data_ptr(), None, None, tg.data_ptr(), 0.7, npix, K,
                                *[o.data_ptr() for o in outs], torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for got, ref in zip(outs, (ra, rl, rd)):
        assert np.abs(n(got).reshape(npix, K) - ref).max() <= TOL * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("K", [7, 40, 130])
def test_composite_every_pair_interacts(hip_lib, K):
    """Windows as wide as the list (small dsd: 3.5 / s spans every depth) and runs of EQUAL depths: every row of a pixel takes a term
    from every column.  The backward's row sums are accumulated by the column walks of the other lanes (composite_core.h,
    compn_bwd_wave<NS, true>: ordered LDS read - add - write) -- here each cell is written once per iteration of every walk."""
    from voge_amd import ops
    rng = np.random.default_rng(100 + K)
    npix = 2 * 41
    ln = np.sort(np.round(rng.uniform(1, 8, (npix, K)) * 4) / 4, axis=1)      # quarter steps: many ties
    act = rng.uniform(0, 4, (npix, K))
    dsd = rng.uniform(0.05, 2.0, (npix, K))
    idx = rng.integers(0, 1000, (npix, K)).astype(np.int32)
    nv = rng.integers(K // 2, K + 1, npix)
    hole = np.arange(K)[None] >= nv[:, None]
    idx[hole], ln[hole], act[hole], dsd[hole] = -1, 1e10, 1e10, 0
    shape = (2, 41, K)
    ta, tl, td = (t(x.reshape(shape), rg=True) for x in (act, ln, dsd))
    w, vn = ops.composite(t(idx.reshape(shape), torch.int32), ta, tl, td, 0.9)
    wr, vr = oracle.composite_fwd(idx, act, ln, dsd, 0.9)
    assert (n(vn).reshape(-1) == vr).all() and np.abs(n(w).reshape(npix, K) - wr).max() < TOL
    gw = rng.normal(size=(npix, K))
    (w * t(gw.reshape(shape))).sum().backward()
    ra, rl, rd = oracle.composite_bwd(act.astype(np.float32), ln.astype(np.float32), dsd.astype(np.float32), gw, 0.9)
    for got, ref, key in ((ta.grad, ra, "g_act"), (tl.grad, rl, "g_len"), (td.grad, rd, "g_dsd")):
        assert np.abs(n(got).reshape(npix, K) - ref).max() <= TOL * max(1.0, np.abs(ref).max()), key


def test_composite_unsorted_list(hip_lib):
    """Lists that are not depth sorted (possible through the public API) take the full K x K scan."""
    from voge_amd import ops
    rng = np.random.default_rng(7)
    npix, K = 2 * 33, 24
    ln = rng.uniform(1, 3, (npix, K))
    ln[::3] = np.sort(ln[::3], axis=1)          # a third of the pixels stay sorted
    act = rng.uniform(0, 5, (npix, K))
    dsd = rng.uniform(1, 400, (npix, K))
    idx = rng.integers(0, 1000, (npix, K)).astype(np.int32)
    shape = (2, 33, K)
    ta, tl, td = (t(x.reshape(shape), rg=True) for x in (act, ln, dsd))
    w, vn = ops.composite(t(idx.reshape(shape), torch.int32), ta, tl, td, 1.3)
    wr, vr = oracle.composite_fwd(idx, act, ln, dsd, 1.3)
    assert np.abs(n(w).reshape(npix, K) - wr).max() < TOL
    gw = rng.normal(size=(npix, K))
    (w * t(gw.reshape(shape))).sum().backward()
    ra, rl, rd = oracle.composite_bwd(act.astype(np.float32), ln.astype(np.float32), dsd.astype(np.float32), gw, 1.3)
    for got, ref in ((ta.grad, ra), (tl.grad, rl), (td.grad, rd)):
        assert np.abs(n(got).reshape(npix, K) - ref).max() <= TOL * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------------------- merge / blend
def test_merge_blend_golden(hip_lib):
    from voge_amd.Renderer import (Fragments, get_silhouette, interpolate_attr, to_colored_background,
                                   to_white_background)
    m = np.load(os.path.join(GOLDEN, "merge_blend.npz"))
    w = t(m["weight"], rg=True)
    colors, feat = t(m["colors"], rg=True), t(m["feat"], rg=True)
    frag = Fragments(vert_weight=w, vert_index=t(m["idx"], torch.int32), valid_num=t(m["valid_num"], torch.int64),
                     vert_hit_length=None)
    rgb = interpolate_attr(frag, colors)
    assert (n(frag.vert_index) == m["idx_after_merge"]).all()      # in-place -1 -> 0 (Aggregation.py:131)
    assert np.abs(n(rgb) - m["rgb"]).max() < TOL
    img = to_white_background(frag, colors)
    assert np.abs(n(img) - m["img_white"]).max() < TOL
    img2 = to_colored_background(frag, colors, background_color=tuple(m["bg"].tolist()), thr=float(m["thr"]))
    assert np.abs(n(img2) - m["img_colored_thr"]).max() < TOL
    assert np.abs(n(get_silhouette(frag)) - m["silhouette"]).max() < TOL
    fm = interpolate_attr(frag, feat)
    assert np.abs(n(fm) - m["feat_map"]).max() < TOL
    ((img * t(m["g_img"])).sum() + (fm * t(m["g_feat"])).sum()).backward()
    assert np.abs(n(w.grad) - m["g_weight"]).max() < TOL * max(1, np.abs(m["g_weight"]).max())
    assert np.abs(n(colors.grad) - m["g_colors"]).max() < TOL * max(1, np.abs(m["g_colors"]).max())
    assert np.abs(n(feat.grad) - m["g_feat_attr"]).max() < TOL * max(1, np.abs(m["g_feat_attr"]).max())


# ------------------------------------------------------------------------------- whole frame
def _render(scene, image_size, B=1, grad=False, rows=None, max_point_per_bin=-1):
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    azims = [scene["azim"] + 25.0 * b for b in range(B)]
    R, T = camera_np.look_at_view_transform([scene["dist"]] * B, [scene["elev"]] * B, azims)
    cams = PerspectiveCameras(focal_length=scene["focal"], principal_point=(scene["principal"],), image_size=(image_size,),
                              device=DEV)
    st = GaussianRenderSettings(image_size=image_size, max_assign=scene["K"], principal=scene["principal"],
                                max_point_per_bin=max_point_per_bin, batch_size=-1)
    renderer = GaussianRenderer(cameras=cams, render_settings=st).to(DEV)
    gm = GaussianMeshes(t(scene["verts"]), t(scene["sigmas"])).to(DEV)
    # a batch of B views indexes attribute rows b*N+n (RayTracing.py:24-30): the table is tiled over the batch
    colors = t(np.tile(scene["colors"], (B, 1)), rg=grad)
    kw = {} if rows is None else dict(rows=rows)
    frag = renderer(gm, R=t(R), T=t(T), **kw)
    img = to_white_background(frag, colors)
    return frag, img, gm, colors, (R, T)


def test_whole_frame_config1_vs_oracle(hip_lib):
    sc = cuboid_scene()
    frag, img, gm, colors, (R, T) = _render(sc, sc["image_size"], grad=True)
    ref = oracle.render(sc["verts"], sc["sigmas"], sc["colors"], R, T, sc["focal"], sc["principal"], sc["image_size"],
                        K=sc["K"])
    same = (n(frag.vert_index) == np.where(ref["idx"] < 0, 0, ref["idx"])).all(-1)   # merge turned -1 into 0
    assert same.mean() > 0.999
    assert (n(frag.valid_num)[same] == ref["valid_num"][same]).all()
    assert np.abs(n(frag.vert_weight)[same] - ref["weight"][same]).max() < TOL
    assert np.abs(n(frag.vert_hit_length)[same] - ref["len"][same])[ref["idx"][same] >= 0].max() < TOL * 10
    assert np.abs(n(img)[same] - ref["image"][same]).max() < TOL
    # flipped pixels change a weight by at most ~thr*e^0.5
    assert np.abs(n(img) - ref["image"]).max() < 0.05
    # backward: image-sum loss, gradients vs the oracle chain
    g_img = np.random.default_rng(0).normal(size=ref["image"].shape)
    (img * t(g_img)).sum().backward()
    K = sc["K"]
    x = ref["rgb"] + (1 - ref["silhouette"])[..., None]
    pass_x = (x < 1).astype(np.float64)
    g_rgb = g_img * pass_x
    g_sil = -(g_rgb.sum(-1)) * (ref["weight"].sum(-1) < 1)
    g_attr, g_w = oracle.merge_bwd(sc["colors"], ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    g_w = g_w + g_sil[..., None]
    g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w, 1.0)
    g_ray, g_mu, g_A = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
    g_sig = 2 * np.einsum("nii->n", g_A)                      # isigma = 2 * s * I
    for name, got, want in (("colors", colors.grad, g_attr), ("verts", gm.verts.grad, g_mu), ("sigmas", gm.sigmas.grad, g_sig)):
        grad_close("whole frame cfg1 " + name, n(got), want, TOL)


def test_row_bands_equal_whole_frame_and_default_bins(hip_lib):
    """Pixel-row tiles (the multi-GPU partition) reproduce the whole frame bit for bit, and the
    default max_point_per_bin=None path equals -1 when nothing is behind the camera."""
    sc = cuboid_scene()
    size = (96, 80)
    sc = dict(sc, focal=100.0, principal=(40.0, 48.0))
    frag, img, *_ = _render(sc, size, B=2)
    parts = [_render(sc, size, B=2, rows=(r0, r1)) for r0, r1 in ((0, 17), (17, 48), (48, 96))]
    for name in ("vert_weight", "vert_index", "valid_num", "vert_hit_length"):
        cat = torch.cat([getattr(p[0], name) for p in parts], dim=1)
        assert torch.equal(cat, getattr(frag, name)), name
    assert torch.equal(torch.cat([p[1] for p in parts], dim=1), img)
    frag2, img2, *_ = _render(sc, size, B=2, max_point_per_bin=None)
    assert torch.equal(img2, img) and torch.equal(frag2.vert_index, frag.vert_index)


def test_striped_rays_fwd_bwd(hip_lib):
    """voge_rays_striped_fwd / _bwd: a rank's interleaved stripes are the same rays as the whole frame's rows (bit for bit),
    and the camera gradients of a loss on them equal the whole-frame kernel's for the same rows."""
    from voge_amd import ops
    from voge_amd.distributed import Stripes
    rng = np.random.default_rng(1)
    R, T = camera_np.look_at_view_transform([3.0, 4.0], [10.0, -20.0], [30.0, 100.0])
    f = np.float32([[300.0, 310.0], [150.0, 140.0]])
    pp = np.float32([[26.0, 18.0], [30.0, 20.5]])
    H, W = 41, 53
    for world, rank, sh in ((3, 1, 4), (2, 0, 8), (4, 3, 5)):
        st = Stripes(H, rank, world, sh)
        rows = st.image_rows(DEV)
        G = t(rng.normal(size=(2, st.h, W, 3)))
        grads = []
        for striped in (True, False):
            tR, tT, tf, tp = t(R, rg=True), t(T, rg=True), t(f, rg=True), t(pp, rg=True)
            if striped:
                rays, origin = ops.pixel_rays(tR, tT, tf, tp, st.row0, st.h, W, st.stripe_h, st.pitch)
            else:
                full, origin = ops.pixel_rays(tR, tT, tf, tp, 0, H, W)
                rays = full.index_select(1, rows)
            ((rays * G).sum() + origin.sum()).backward()
            grads.append((n(rays), [n(x.grad) for x in (tR, tT, tf, tp)]))
        assert np.array_equal(grads[0][0], grads[1][0])
        for a, b in zip(grads[0][1], grads[1][1]):
            assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max())


def test_interleaved_stripes_equal_whole_frame(hip_lib):
    """One frame dealt to the ranks in interleaved stripes (distributed.Stripes, voge_rays_striped_fwd): every rank's
    stripes, stacked into one image and rendered by ONE renderer call, reproduce their rows of the whole frame bit for bit
    (pixels are independent; the binning's cones bound whatever rays are present), for stripe heights that are and are not
    multiples of the 32-row super-tiles, with a cut last stripe, B = 2."""
    from voge_amd.distributed import Stripes
    sc = cuboid_scene()
    size = (100, 80)
    sc = dict(sc, focal=100.0, principal=(40.0, 50.0))
    frag, img, *_ = _render(sc, size, B=2)
    for world, stripe_h in ((3, 32), (2, 8), (4, 5)):
        seen = []
        for rank in range(world):
            st = Stripes(size[0], rank, world, stripe_h)
            rows = st.image_rows(DEV)
            seen += rows.tolist()
            f, im, *_ = _render(sc, size, B=2, rows=st)
            assert im.shape[1] == st.h
            for name in ("vert_weight", "vert_index", "valid_num", "vert_hit_length"):
                assert torch.equal(getattr(f, name), getattr(frag, name).index_select(1, rows)), (name, world, stripe_h, rank)
            assert torch.equal(im, img.index_select(1, rows))
        assert sorted(seen) == list(range(size[0]))


def test_full_size_properties_config3(hip_lib):
    """BASELINE config 3 (50k Gaussians, 512x512, K=40): too large for the oracle in seconds, so
    check size-independent properties + an oracle spot check on a row band."""
    verts, sig, colors = random_scene(50000, seed=0)
    sc = dict(verts=verts, sigmas=sig, colors=colors, focal=600.0, principal=(256.0, 256.0), image_size=(512, 512),
              dist=4.0, elev=10.0, azim=70.0, K=40)
    frag, img, gm, cols, (R, T) = _render(sc, (512, 512), grad=True)
    idx, w, vn, hl = n(frag.vert_index), n(frag.vert_weight), n(frag.valid_num), n(frag.vert_hit_length)
    K = 40
    slot = np.arange(K)[None, None, None]
    filled = slot < vn[..., None]
    assert np.isfinite(w).all() and (w >= 0).all() and (w[~filled] == 0).all()
    assert (hl[~filled] == np.float32(1e10)).all()
    d = np.diff(hl, axis=-1)
    assert (d[filled[..., 1:]] >= 0).all(), "hit lengths must ascend within the valid prefix"
    assert (idx[filled] >= 0).all() and (idx[filled] < 50000).all()
    srt = np.sort(np.where(filled, idx, -1 - slot), axis=-1)
    assert (np.diff(srt, axis=-1) != 0).all(), "a Gaussian may appear once per pixel"
    im = n(img)
    assert im.min() >= 0 and im.max() <= 1 and (im[0, 0, 0] == 1).all() and vn.max() == K
    # determinism of the forward
    frag_b, img_b, *_ = _render(sc, (512, 512))
    assert torch.equal(img_b, img.detach()) and torch.equal(frag_b.vert_weight, frag.vert_weight.detach())
    # oracle spot check on rows 250..253
    rays, origin = camera_np.pixel_rays(R, T, 600.0, (256.0, 256.0), (512, 512))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(0.01)
    ref = oracle.trace_fwd(mus, isg, rays[:, 250:254], K, thr_act)
    wr, vr = oracle.composite_fwd(*[ref[i] for i in (0, 2, 1, 3)], 1.0)
    same = (np.where(ref[0] < 0, 0, ref[0]) == idx[:, 250:254]).all(-1)
    assert same.mean() > 0.995
    assert np.abs(w[:, 250:254][same] - wr[same]).max() < TOL
    # backward runs and is finite; colour gradient equals the sum of weights per Gaussian
    img.sum().backward()
    assert torch.isfinite(gm.verts.grad).all() and torch.isfinite(gm.sigmas.grad).all()
    assert gm.verts.grad.abs().max() > 0


# ------------------------------------------------------------------------------- ray generation
def _rays_torch64(R, T, f, pp, H, W):
    """fp64 torch statement of the a-0 convention (autograd reference for the ray kernel)."""
    ii = torch.arange(H, dtype=torch.float64) + 0.5
    jj = torch.arange(W, dtype=torch.float64) + 0.5
    B = R.shape[0]
    x = (pp[:, 0, None] - jj[None]) / f[:, 0, None]
    y = (pp[:, 1, None] - ii[None]) / f[:, 1, None]
    dv = torch.stack([x[:, None, :].expand(B, H, W), y[:, :, None].expand(B, H, W), torch.ones(B, H, W, dtype=torch.float64)], -1)
    Rinv = torch.linalg.inv(R)
    dw = torch.einsum("bhwj,bjk->bhwk", dv, Rinv)
    return dw / dw.norm(dim=-1, keepdim=True), -torch.einsum("bj,bjk->bk", T, Rinv)


def test_pixel_rays_fwd_bwd(hip_lib):
    from voge_amd import ops
    rng = np.random.default_rng(0)
    R, T = camera_np.look_at_view_transform([3.0, 4.0], [10.0, -20.0], [30.0, 100.0])
    R = (R + rng.normal(size=R.shape) * 0.01).astype(np.float32)          # not exactly orthonormal
    f = np.float32([[300.0, 310.0], [150.0, 140.0]])
    pp = np.float32([[26.0, 18.0], [30.0, 20.5]])
    H, W = 37, 53
    ref_rays, ref_origin = camera_np.pixel_rays(R, T, f, pp, (H, W))
    tR, tT, tf, tp = t(R, rg=True), t(T, rg=True), t(f, rg=True), t(pp, rg=True)
    rays, origin = ops.pixel_rays(tR, tT, tf, tp, 0, H, W)
    assert np.abs(n(rays) - ref_rays).max() < 2e-6 and np.abs(n(origin) - ref_origin).max() < 1e-5
    band, _ = ops.pixel_rays(tR, tT, tf, tp, 10, 7, W)
    assert torch.equal(band, rays[:, 10:17])
    G, Go = rng.normal(size=ref_rays.shape), rng.normal(size=(2, 3))
    ((rays * t(G)).sum() + (origin * t(Go)).sum()).backward()
    c = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    cR, cT, cf, cp = c(R), c(T), c(f), c(pp)
    r64, o64 = _rays_torch64(cR, cT, cf, cp, H, W)
    ((r64 * torch.tensor(G)).sum() + (o64 * torch.tensor(Go)).sum()).backward()
    for name, got, want in (("R", tR.grad, cR.grad), ("T", tT.grad, cT.grad), ("focal", tf.grad, cf.grad), ("pp", tp.grad, cp.grad)):
        w = want.numpy()
        assert np.abs(n(got) - w).max() <= 2e-4 * max(1.0, np.abs(w).max()), name


def test_camera_pose_gradient_end_to_end(hip_lib):
    """Gradients reach R and T through the rays, the camera-centred means and the trace backward
    (what the pose-estimation use of the renderer needs).  Reference: the oracle's backward chain
    down to (g_ray, g_mu), then fp64 autograd through the a-0 camera convention."""
    sc = cuboid_scene()
    size = (48, 48)
    sc = dict(sc, focal=56.0, principal=(24.0, 24.0), K=12)
    from voge_amd.Meshes import GaussianMeshesNaive
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    R0, T0 = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
    cams = PerspectiveCameras(focal_length=sc["focal"], principal_point=(sc["principal"],), image_size=(size,), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=size, max_assign=sc["K"], max_point_per_bin=-1))
    gm = GaussianMeshesNaive(t(sc["verts"]), t(sc["sigmas"]))
    colors = t(sc["colors"])
    g_img = np.random.default_rng(1).normal(size=(1,) + size + (3,))
    Rv, Tv = t(R0, rg=True), t(T0, rg=True)
    img = to_white_background(renderer(gm, R=Rv, T=Tv), colors)
    (img * t(g_img)).sum().backward()

    ref = oracle.render(sc["verts"], sc["sigmas"], sc["colors"], R0, T0, sc["focal"], sc["principal"], size, K=sc["K"])
    assert np.abs(n(img) - ref["image"]).max() < 1e-3
    x = ref["rgb"] + (1 - ref["silhouette"])[..., None]
    g_rgb = g_img * (x < 1)
    g_sil = -(g_rgb.sum(-1)) * (ref["weight"].sum(-1) < 1)
    _, g_w = oracle.merge_bwd(sc["colors"], ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w + g_sil[..., None], 1.0)
    g_ray, g_mu, _ = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
    c = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    cR, cT = c(R0), c(T0)
    f64 = torch.tensor([[sc["focal"]] * 2], dtype=torch.float64)
    p64 = torch.tensor([list(sc["principal"])], dtype=torch.float64)
    r64, o64 = _rays_torch64(cR, cT, f64, p64, *size)
    ((r64 * torch.tensor(g_ray)).sum() + (o64 * torch.tensor(-g_mu.sum(0, keepdims=True))).sum()).backward()
    for name, got, want in (("R", Rv.grad, cR.grad), ("T", Tv.grad, cT.grad)):
        grad_close("camera pose " + name, n(got), want.numpy(), TOL)


# ------------------------------------------------------------------------------- row sharding, 2 processes
def _shard_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share cuda:0 here; RCCL needs one GPU per rank
    from voge_amd.distributed import allreduce_grads, gather_rows, row_band
    sc = cuboid_scene()
    size = (75, 64)
    sc = dict(sc, focal=80.0, principal=(32.0, 37.0))
    r0, r1 = row_band(size[0], rank, world)
    frag, band, gm, colors, _ = _render(sc, size, grad=True, rows=(r0, r1))
    img = gather_rows(band, size[0])
    g = torch.linspace(0.5, 1.5, img.numel(), device=DEV).view_as(img)
    (img[:, r0:r1] * g[:, r0:r1]).sum().backward()
    allreduce_grads([gm.verts, gm.sigmas, colors])
    if rank == 0:
        torch.save({"img": img.detach().cpu(), "gv": gm.verts.grad.cpu(), "gs": gm.sigmas.grad.cpu(), "gc": colors.grad.cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_sharding_matches_single_process(hip_lib, tmp_path):
    """The multi-GPU path (pixel-row bands + one all_gather + one all_reduce), exercised with two
    processes on this box's single GPU: forward identical, gradients within atomics tolerance."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_shard_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    sc = cuboid_scene()
    size = (75, 64)
    sc = dict(sc, focal=80.0, principal=(32.0, 37.0))
    frag, img, gm, colors, _ = _render(sc, size, grad=True)
    g = torch.linspace(0.5, 1.5, img.numel(), device=DEV).view_as(img)
    (img * g).sum().backward()
    assert torch.equal(got["img"], img.detach().cpu())
    for a, b in ((got["gv"], gm.verts.grad), (got["gs"], gm.sigmas.grad), (got["gc"], colors.grad)):
        b = b.cpu()
        assert (a - b).abs().max() <= 1e-4 * max(1.0, b.abs().max().item())


def _stacked_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share cuda:0 here
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    from voge_amd.distributed import allreduce_grads, gather_stacked, render_stacked, stacked_bounds
    sc = dict(cuboid_scene(), focal=80.0, principal=(32.0, 24.0))
    B, (H, W) = 3, (48, 64)
    R, T = camera_np.look_at_view_transform([sc["dist"]] * B, [sc["elev"]] * B, [sc["azim"] + 25.0 * b for b in range(B)])
    R, T = t(R), t(T)
    cams = PerspectiveCameras(focal_length=sc["focal"], principal_point=(sc["principal"],), image_size=((H, W),), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=sc["K"], max_point_per_bin=-1)).to(DEV)
    gm = GaussianMeshes(t(sc["verts"]), t(sc["sigmas"])).to(DEV)
    colors = t(sc["colors"], rg=True)

    def render_views(b0, b1, r0, r1):
        frag = renderer(gm, R=R[b0:b1], T=T[b0:b1], rows=(r0, r1))
        return to_white_background(frag, colors.repeat(b1 - b0, 1))
    bounds = stacked_bounds(B, H, world)           # 3 views on 2 ranks: the cut falls in the middle of view 1
    rows = render_stacked(render_views, bounds[rank], bounds[rank + 1], H)
    img = gather_stacked(rows, B, H)
    g = torch.linspace(0.5, 1.5, img.numel(), device=DEV).view_as(img)
    s0, s1 = bounds[rank], bounds[rank + 1]
    (img.reshape(B * H, W, 3)[s0:s1] * g.reshape(B * H, W, 3)[s0:s1]).sum().backward()
    allreduce_grads([gm.verts, gm.sigmas, colors])
    if rank == 0:
        torch.save({"img": img.detach().cpu(), "gv": gm.verts.grad.cpu(), "gs": gm.sigmas.grad.cpu(), "gc": colors.grad.cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stacked_view_row_sharding_matches_single_process(hip_lib, tmp_path):
    """A batch of three views over two ranks on the stacked (view, row) axis (distributed.stacked_bounds: view first,
    row bands only where a cut falls inside a view), two processes on this box's single GPU: the gathered batch is
    identical to the one-call render, the all-reduced gradients agree within atomics tolerance."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_stacked_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    sc = dict(cuboid_scene(), focal=80.0, principal=(32.0, 24.0))
    frag, img, gm, colors, _ = _render(sc, (48, 64), B=3, grad=True)
    g = torch.linspace(0.5, 1.5, img.numel(), device=DEV).view_as(img)
    (img * g).sum().backward()
    assert torch.equal(got["img"], img.detach().cpu())
    N = sc["verts"].shape[0]
    for a, b in ((got["gv"], gm.verts.grad), (got["gs"], gm.sigmas.grad), (got["gc"], colors.grad.view(3, N, 3).sum(0))):
        b = b.cpu()
        assert (a - b).abs().max() <= 1e-4 * max(1.0, b.abs().max().item())


def test_shape_fitting_loop_converges(hip_lib):
    """BASELINE config 5 / demo/ShapeFitting.py:250-296: multi-view SGD on the vertices and colours of a
    Gaussian ico-sphere against silhouette (+ rgb) targets.  Reduced size (642 Gaussians, 64x64, 160
    iterations, rgb from iteration 40).  The reference's targets come from PyTorch3D's mesh rasteriser,
    so there are no reference numbers: the loop is pinned by both losses going down."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("shape_fitting_demo", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "demo", "ShapeFitting.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    v, f = demo.ico_sphere(4)
    assert v.shape == (2562, 3) and f.shape == (5120, 3)                      # what ShapeFitting.py:216 builds
    h = demo.fit(iters=160, level=3, size=64, max_assign=16, rgb_on=40, quiet=True)
    sil, rgb = np.asarray(h["silhouette"]), np.asarray(h["rgb"])
    assert np.isfinite(sil).all() and np.isfinite(rgb).all()
    assert sil[-20:].mean() < 0.35 * sil[:5].mean(), (sil[:5].mean(), sil[-20:].mean())
    assert rgb[-20:].mean() < 0.6 * rgb[40:45].mean(), (rgb[40:45].mean(), rgb[-20:].mean())


@pytest.mark.parametrize("inverse_sigma", [False, True])
def test_fused_preamble_equals_reference_ops(hip_lib, inverse_sigma):
    """Renderer.py:130-137 (`verts - origin`, `2 * sigmas` or `2 / sigmas`) folded into the trace kernels
    (voge_trace_topk_fwd_iso_view / voge_trace_bwd_iso_view) against the same renderer running those lines
    as torch ops: identical fragments, same gradients (two views share one Gaussian set: the backward
    sums over the batch)."""
    import voge_amd.Renderer as Rm
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    N, H, W, K = 1500, 72, 88, 12
    verts, sig, cols = random_scene(N, seed=21, lo=0.04, hi=0.12)
    sig = np.ascontiguousarray(sig[:, 0, 0]) if sig.ndim == 3 else sig
    if inverse_sigma:
        sig = (1.0 / sig).astype(np.float32)
    R, T = look_at_view_transform(dist=[3.0, 3.4], elev=[10.0, -20.0], azim=[30.0, 200.0], device="cuda")
    cams = PerspectiveCameras(focal_length=90.0, principal_point=((W / 2.0, H / 2.0),), image_size=((H, W),), device="cuda")
    st = Rm.GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1.2,
                                   inverse_sigma=inverse_sigma, max_point_per_bin=-1)
    renderer = Rm.GaussianRenderer(cams, st).to("cuda")
    g_img = torch.randn(2, H, W, 3, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    out = {}
    for fused in (True, False):
        Rm.FUSED_PREAMBLE = fused
        try:
            gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to("cuda")
            colors = t(np.tile(cols, (2, 1)), rg=True)          # two views: attribute rows b*N+n
            frag = renderer(gm, R=R, T=T)
            img = Rm.to_white_background(frag, colors)
            (img * g_img).sum().backward()
            out[fused] = (n(frag.vert_index), n(frag.vert_weight), n(frag.vert_hit_length), n(img),
                          n(gm.verts.grad), n(gm.sigmas.grad), n(colors.grad))
        finally:
            Rm.FUSED_PREAMBLE = True
    a, b = out[True], out[False]
    assert (a[0] == b[0]).all()
    for x, y in zip(a[1:4], b[1:4]):
        assert np.array_equal(x, y)                      # same arithmetic: bit-identical forward
    assert (a[1] > 0).mean() > 0.05
    for name, x, y in zip(("verts", "sigmas", "colors"), a[4:], b[4:]):
        scale = max(1.0, float(np.abs(y).max()))
        assert np.abs(x - y).max() <= 2e-5 * scale, name
    assert np.abs(b[5]).max() > 0


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_iso_view_entry_points_batched_inputs(hip_lib, mode):
    """voge_trace_topk_fwd_iso_view / voge_trace_bwd_iso_view with per-view Gaussian sets ([B,N,3] verts,
    [B,N] sigmas: shared = 0) and every sigma mode, against the plain isotropic calls on inputs prepared with
    torch ops (verts - origin, a = sigma | 2 sigma | 2 / sigma)."""
    from voge_amd import ops
    rng = np.random.default_rng(31 + mode)
    B, N, H, W, K = 2, 700, 40, 56, 9
    verts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    sig = rng.uniform(60, 250, (B, N)).astype(np.float32)
    if mode == 2:
        sig = (1.0 / sig).astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.2] * B, [12.0] * B, [40.0, 75.0])
    rays, origin = camera_np.pixel_rays(R, T, 60.0, (W / 2.0, H / 2.0), (H, W))
    rays, origin = rays.astype(np.float32), origin.astype(np.float32)
    thr_act = oracle.thr_act_of(0.01)
    g = [torch.randn(B, H, W, K, device="cuda", generator=torch.Generator("cuda").manual_seed(5 + i)) for i in range(3)]
    # fused
    v1, s1 = t(verts, rg=True), t(sig, rg=True)
    out1 = ops._RayTraceVoGEIsoView.apply(v1, s1, t(origin), t(rays), None, thr_act, K, mode)
    sum(((o * gi).sum() for o, gi in zip(out1[1:], g))).backward()
    # reference composition
    v2, s2 = t(verts, rg=True), t(sig, rg=True)
    a = s2 if mode == 0 else (2.0 * s2 if mode == 1 else 2.0 / s2)
    out2 = ops._RayTraceVoGEIso.apply((v2 - t(origin)[:, None]).reshape(-1, 3), a.reshape(-1), t(rays), None, thr_act, K)
    sum(((o * gi).sum() for o, gi in zip(out2[1:], g))).backward()
    assert (n(out1[0]) == n(out2[0])).all() and (n(out1[0]) >= 0).mean() > 0.05
    for x, y in zip(out1[1:], out2[1:]):
        assert torch.equal(x, y)
    for name, x, y in (("verts", v1.grad, v2.grad), ("sigmas", s1.grad, s2.grad)):
        scale = max(1.0, float(y.abs().max()))
        assert float((x - y).abs().max()) <= 2e-5 * scale, name


# ------------------------------------------------------------------------------- bookkeeping on the index tensor
def test_list_path_backward_after_merge_and_index_guards(hip_lib):
    """(a) Explicit bin lists: merge_final rewrites -1 -> 0 in the fragments' index tensor (Aggregation.py:131); a
    later backward with a gradient on an EMPTY slot (e.g. a loss on vert_hit_length) must not land on Gaussian 0 --
    the list kernel's per-pixel hit count marks the filled slots.  (b) A batch of views indexes rows b*N+n: an
    attribute table with N rows raises like the reference's assert (:120).  (c) Editing sel_idx through torch between
    ray_tracing and aggregation invalidates the cached hit count (valid_num = sum(sel_idx >= 0), :104)."""
    from voge_amd import ops
    from voge_amd.Aggregation import aggregation, merge_final
    verts, sig, cols = random_scene(300, seed=3, lo=0.05, hi=0.1)
    H, W, K, bs = 24, 32, 12, 10
    sc = dict(verts=verts, sigmas=sig, focal=30.0, principal=(16.0, 12.0), image_size=(H, W), dist=3.0, elev=5.0, azim=15.0)
    mus, isg, rays, _, _ = camera_inputs(sc, B=2)
    BH, BW = (H - 1) // bs + 1, (W - 1) // bs + 1
    bins = np.broadcast_to(np.arange(300, dtype=np.int32)[None, None, None], (2, BH, BW, 300)).copy()
    bins[1] += 300
    thr_act = oracle.thr_act_of(0.01)
    tm, tA = t(mus.reshape(-1, 3), rg=True), t(isg.reshape(-1, 3, 3), rg=True)
    idx, ln, act, dsd = ops.ray_trace_fine(tm, tA, t(rays), t(bins, torch.int32), thr_act, bs, K)
    empty = n(idx) < 0
    assert empty.any() and (~empty).any()
    w, idx2, vn, hl = aggregation(idx, act, ln, dsd, 1.0)
    with pytest.raises(AssertionError):
        merge_final(t(cols), w, idx2, vn)                        # (b) 300 rows, indices up to 599
    rgb = merge_final(t(np.tile(cols, (2, 1))), w, idx2, vn)
    assert (n(idx2)[empty] == 0).all()                           # the in-place fix has run
    g_hl = np.ones(idx.shape) * 3.0                              # gradient on every slot, empty ones included
    (rgb.sum() + (hl * t(g_hl)).sum()).backward()
    # oracle: the same loss with the gradient of the empty slots masked (sentinels carry no gradient)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act, bin_points=bins, bin_size=bs)
    wr, vr = oracle.composite_fwd(ref[0], ref[2], ref[1], ref[3], 1.0)
    _, g_w = oracle.merge_bwd(np.tile(cols, (2, 1)), ref[0], wr, vr, np.ones(rgb.shape))
    g_act, g_len, g_dsd = oracle.composite_bwd(ref[2], ref[1], ref[3], g_w, 1.0)
    _, g_mu, g_A = oracle.trace_bwd(mus, isg, rays, ref[0], g_len + g_hl * (ref[0] >= 0), g_act, g_dsd)
    grad_close("list path after merge, g_mu", n(tm.grad), g_mu, TOL)
    assert np.abs(n(tm.grad)[0] - g_mu[0]).max() <= TOL * max(1.0, np.abs(g_mu).max())      # Gaussian 0 in particular
    # (c) mask a Gaussian out of the fragments by hand: the count must follow
    idx3, ln3, act3, dsd3 = ops.ray_trace_fine(t(mus.reshape(-1, 3)), t(isg.reshape(-1, 3, 3)), t(rays), None, thr_act, bs, K)
    assert ops.hit_count_of(idx3) is not None
    first = n(idx3)[..., 0] >= 0
    idx3[..., 0] = -1
    assert ops.hit_count_of(idx3) is None
    _, _, vn3, _ = aggregation(idx3, act3, ln3, dsd3, 1.0)
    assert (n(vn3) == (n(idx3) >= 0).sum(-1)).all() and first.any()


# ------------------------------------------------------------------------------- fused trace + composite
@pytest.mark.parametrize("mode,K", [(2, 40), (1, 12), (0, 24), (0, 10), (2, 7)])
def test_fused_fragments_equal_trace_then_composite(hip_lib, mode, K):
    """voge_fragments_fwd* (the sweep composites in its epilogue when K % 4 == 0, else runs the composite kernel
    behind the trace) against the two stand-alone entry points called one after the other: identical fragments,
    bit for bit, and the same gradients (VoGE/Renderer.py:139-150 = ray_tracing + aggregation)."""
    from voge_amd import ops
    from voge_amd.Aggregation import aggregation
    B, N, H, W = 2, 3000, 72, 88
    verts, sig, _ = random_scene(N, seed=40 + K, aniso=(mode == 0), lo=0.04, hi=0.1)
    if mode == 0:
        sig[::2] = np.eye(3, dtype=np.float32)[None] * sig[::2, :1, :1]           # a mix of isotropic and full forms
    R, T = camera_np.look_at_view_transform([3.0, 3.3], [10.0, -15.0], [30.0, 120.0])
    rays_np, origin = camera_np.pixel_rays(R, T, 90.0, (W / 2.0, H / 2.0), (H, W))
    from voge_amd.cameras import PerspectiveCameras, pixel_rays
    cams = PerspectiveCameras(focal_length=90.0, principal_point=((W / 2.0, H / 2.0),), image_size=((H, W),), R=R, T=T, device=DEV)
    thr_act = oracle.thr_act_of(0.01)
    g_w = torch.randn(B, H, W, K, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    g_l = torch.randn(B, H, W, K, device=DEV, generator=torch.Generator(DEV).manual_seed(4)) * 0.1
    out = {}
    for fused in (True, False):
        rays, org = pixel_rays(cams, (H, W))
        assert ops.cones_of(rays, B, H, W) is not None
        if mode == 2:
            p0, p1 = t(verts, rg=True), t(sig, rg=True)
            args = (p0, p1, org)
        else:
            mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32).reshape(-1, 3)
            p0 = t(mus, rg=True)
            p1 = t(np.tile(2 * sig, (B,) + (1,) * (sig.ndim - 1)), rg=True)
            args = (p0, p1, None)
        if fused:
            w, idx, vn, hl = ops.fragments(mode, args[0], args[1], args[2], rays, None, thr_act, K, 1 if mode == 2 else 0, 1.2)
        else:
            if mode == 2:
                sel = ops._RayTraceVoGEIsoView.apply(p0, p1, org, rays, None, thr_act, K, 1)
            elif mode == 1:
                sel = ops._RayTraceVoGEIso.apply(p0, p1, rays, None, thr_act, K)
            else:
                sel = ops.ray_trace_fine(p0, p1, rays, None, thr_act, 10, K)
            w, idx, vn, hl = aggregation(sel[0], sel[2], sel[1], sel[3], 1.2)
        ((w * g_w).sum() + (torch.where(idx >= 0, hl, torch.zeros_like(hl)) * g_l).sum()).backward()
        out[fused] = [n(x) for x in (w, idx, vn, hl, p0.grad, p1.grad)]
    a, b_ = out[True], out[False]
    assert (a[1] >= 0).mean() > 0.05 and a[2].max() == K
    for x, y, name in zip(a[:4], b_[:4], ("weight", "idx", "valid_num", "hit_len")):
        assert np.array_equal(x, y), name
    for x, y, name in zip(a[4:], b_[4:], ("d/d means", "d/d sigmas")):
        assert np.abs(x - y).max() <= 2e-5 * max(1.0, np.abs(y).max()), name


@pytest.mark.parametrize("case", ["a", "b"])
def test_renderer_against_the_running_reference_host_logic(hip_lib, case):
    """tests/golden/host_logic.npz (the imported reference's GaussianRenderer.forward + ray_tracing run with a recording
    stand-in for its CUDA kernel, answered by the oracle): this renderer, given the same (verts, sigmas, R, T, camera,
    settings), returns the same Fragments -- scalar sigmas, and full covariances with inverse_sigma=True on two views."""
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings
    from voge_amd.cameras import PerspectiveCameras
    g = np.load(os.path.join(GOLDEN, "host_logic.npz"))
    c = case
    H, W = (int(x) for x in g[c + "_size"])
    cams = PerspectiveCameras(focal_length=float(g[c + "_focal"]), principal_point=(tuple(g[c + "_pp"].tolist()),),
                              image_size=((H, W),), device=DEV)
    st = GaussianRenderSettings(image_size=(H, W), max_assign=int(g[c + "_K"]), thr_activation=float(g[c + "_thr"]),
                                absorptivity=float(g[c + "_occ"]), inverse_sigma=bool(g[c + "_inverse_sigma"]), max_point_per_bin=-1)
    renderer = GaussianRenderer(cams, st).to(DEV)
    gm = GaussianMeshes(t(g[c + "_verts"]), t(g[c + "_sigmas"])).to(DEV)
    frag = renderer(gm, R=t(g[c + "_R"]), T=t(g[c + "_T"]))
    idx = n(frag.vert_index)
    same = (idx == g[c + "_index"]).all(-1)
    from util import _report_flips
    _report_flips(f"host logic {case}", (~same).sum(), same.size)
    assert (~same).sum() <= 0.004 * same.size            # the reference's kernel stand-in computes in fp32: boundary flips
    assert (n(frag.valid_num)[same] == g[c + "_valid_num"][same]).all()
    e_w = np.abs(n(frag.vert_weight)[same] - g[c + "_weight"][same]).max()
    hit = (g[c + "_index"] >= 0) & same[..., None]
    e_l = np.abs(n(frag.vert_hit_length)[hit] - g[c + "_hit_length"][hit]).max()
    from util import log_line
    log_line(f"[parity] host logic {case}: max weight error {e_w:.2e}, max hit length error {e_l:.2e} (fixture: the fp32 stand-in)")
    assert e_w < 3e-4 and e_l < 4e-4


@pytest.mark.parametrize("K,B,inverse,aniso", [(40, 1, False, False), (12, 2, True, False), (26, 1, False, False),
                                               (20, 1, False, True), (8, 2, False, True)])
def test_fused_fragment_backward_equals_the_three_kernels(hip_lib, K, B, inverse, aniso, monkeypatch):
    """voge_fragment_shade_bwd_iso (shade -> composite -> trace backward in one kernel, taken by to_colored_background
    on this renderer's fragments) against the three stand-alone backward kernels on the same frame -- with a second
    consumer of the weights (a silhouette loss, which keeps flowing through _Fragments.backward) and a loss on
    vert_hit_length, whose gradients must simply add up."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, to_colored_background
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    N, H, W = 2500, 70, 90
    verts, sig, cols = random_scene(N, seed=60 + K, lo=0.04, hi=0.1, aniso=aniso)      # aniso: [N,3,3] forms, general kernels
    if inverse:
        sig = (1.0 / sig).astype(np.float32)
    R, T = look_at_view_transform(dist=[3.0, 3.4][:B], elev=[10.0, -20.0][:B], azim=[30.0, 200.0][:B], device=DEV)
    cams = PerspectiveCameras(focal_length=95.0, principal_point=((W / 2.0, H / 2.0),), image_size=((H, W),), device=DEV)
    st = GaussianRenderSettings(image_size=(H, W), max_assign=K, absorptivity=1.1, inverse_sigma=inverse, max_point_per_bin=-1)
    renderer = GaussianRenderer(cams, st).to(DEV)
    gen = torch.Generator(DEV).manual_seed(9)
    g_img = torch.randn(B, H, W, 3, device=DEV, generator=gen)
    g_sil = torch.randn(B, H, W, device=DEV, generator=gen)
    out = {}
    for fused in (True, False):
        # (fused: the renderer's default -- deferred composite, one-pass shade forward, one-kernel backward;
        #  not fused: the eager chain with every stage a kernel of its own)
        monkeypatch.setattr(ops, "LAZY_COMPOSITE", fused)
        gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(DEV)
        colors = t(np.tile(cols, (B, 1)), rg=True)
        frag = renderer(gm, R=R, T=T)
        if not fused:
            assert hasattr(frag.vert_weight, "voge_through")
        if not fused:
            del frag.vert_weight.voge_through
        # the comparison chain: voge_shade_bwd -> voge_composite_bwd -> voge_trace_bwd*, act / dsd materialised
        monkeypatch.setattr(ops, "THREE_KERNEL_BACKWARD", not fused)
        img = to_colored_background(frag, colors, background_color=(0.9, 0.8, 1.0))
        assert (type(img.grad_fn).__name__ in ("_ShadeThroughBackward", "_CompositeShadeBackward")) == fused
        hl = torch.where(frag.vert_index >= 0, frag.vert_hit_length, torch.zeros_like(frag.vert_hit_length))
        ((img * g_img).sum() + (get_silhouette(frag) * g_sil).sum() + 0.01 * hl.sum()).backward()
        out[fused] = [n(x) for x in (img, gm.verts.grad, gm.sigmas.grad, colors.grad)]
    assert np.abs(out[True][0] - out[False][0]).max() <= 1e-6      # (one pass sums per lane group, the shade kernel per DPP row)
    for name, x, y in zip(("verts", "sigmas", "colors"), out[True][1:], out[False][1:]):
        scale = max(1.0, float(np.abs(y).max()))
        assert np.abs(x - y).max() <= 3e-5 * scale, (name, float(np.abs(x - y).max()), scale)
    assert np.abs(out[False][1]).max() > 0 and np.abs(out[False][2]).max() > 0


def test_fused_fragment_backward_reads_a_broadcast_gradient_in_place(hip_lib):
    """sum() / mean() losses hand the fused backward ONE scalar broadcast over the image (all strides zero); it is read in
    place (g_stride_pix = g_stride_c = 0) and must give what the materialised gradient gives."""
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    N, H, W, K = 3000, 64, 80, 20
    verts, sig, cols = random_scene(N, seed=77, lo=0.04, hi=0.1)
    R, T = look_at_view_transform(dist=3.0, elev=10.0, azim=30.0, device=DEV)
    cams = PerspectiveCameras(focal_length=95.0, principal_point=((W / 2.0, H / 2.0),), image_size=((H, W),), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(DEV)
    out = []
    for broadcast in (True, False):
        gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(DEV)
        colors = t(cols, rg=True)
        img = to_white_background(renderer(gm, R=R, T=T), colors)
        assert type(img.grad_fn).__name__ in ("_ShadeThroughBackward", "_CompositeShadeBackward")
        if broadcast:
            (0.5 * img.mean()).backward()
        else:
            img.backward(torch.full_like(img, 0.5 / img.numel()))
        out.append([n(x) for x in (gm.verts.grad, gm.sigmas.grad, colors.grad)])
    for name, x, y in zip(("verts", "sigmas", "colors"), out[0], out[1]):
        scale = max(1e-6, float(np.abs(y).max()))
        assert np.abs(x - y).max() <= 2e-5 * scale, (name, float(np.abs(x - y).max()), scale)
    assert np.abs(out[1][0]).max() > 0


def test_fragments_without_act_dsd_equal_the_kept_form(hip_lib, monkeypatch):
    """Scalar-sigma fragments keep no act / dsd by default (voge_fragments_fwd_iso* with NULL for both; composite and
    fused backward derive them from the records).  Against VOGE_FRAGMENTS_KEEP_ACT_DSD=1 (the reference's layout):
    identical fragments bit for bit, act / dsd materialised on request (voge_fragment_act_dsd_iso) identical to the kept
    arrays wherever a slot is live, gradients of both backward routes within tolerance."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, to_white_background
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    N, H, W, K = 3000, 72, 88, 24
    verts, sig, cols = random_scene(N, seed=91, lo=0.04, hi=0.1)
    R, T = look_at_view_transform(dist=[3.0, 3.3], elev=[10.0, -15.0], azim=[30.0, 160.0], device=DEV)
    cams = PerspectiveCameras(focal_length=95.0, principal_point=((W / 2.0, H / 2.0),), image_size=((H, W),), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(DEV)
    gen = torch.Generator(DEV).manual_seed(3)
    g_sil = torch.randn(2, H, W, device=DEV, generator=gen)
    out = {}
    for keep in ("1", "0"):
        monkeypatch.setenv("VOGE_FRAGMENTS_KEEP_ACT_DSD", keep)
        gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(DEV)
        colors = t(np.tile(cols, (2, 1)), rg=True)
        frag = renderer(gm, R=R, T=T)
        th = frag.vert_weight.voge_through
        assert (th["act"] is None) == (keep == "0")
        img = to_white_background(frag, colors)
        # the image through the fused backward, the silhouette through _Fragments.backward (which asks for act / dsd)
        (img.sum() + (get_silhouette(frag) * g_sil).sum()).backward()
        act, dsd = ops._act_dsd([th["act"], th["dsd"]], th["records"], th["rays"], th["idx"], th["len"], th["cnt"], th["B"] * th["N"])
        out[keep] = [n(x) for x in (frag.vert_weight, frag.vert_index, frag.vert_hit_length, frag.valid_num, img, act, dsd,
                                    gm.verts.grad, gm.sigmas.grad, colors.grad)]
    for a, b in zip(out["1"][:5], out["0"][:5]):
        assert np.array_equal(a, b)
    live = np.arange(K)[None, None, None, :] < out["1"][3][..., None]
    assert live.any() and np.array_equal(out["1"][5][live], out["0"][5][live]) and np.array_equal(out["1"][6][live], out["0"][6][live])
    for name, a, b in zip(("verts", "sigmas", "colors"), out["1"][7:], out["0"][7:]):
        grad_close("lean vs kept act/dsd, " + name, b, a, TOL)


@pytest.mark.parametrize("N,H,W,K,B,aniso,fused", [
    (0, 16, 16, 8, 1, False, True), (1, 1, 1, 1, 1, False, True), (5, 3, 7, 3, 1, False, True),
    (300, 33, 47, 7, 1, False, True), (300, 33, 47, 8, 2, False, True), (300, 20, 20, 130, 1, False, False),
    (300, 20, 20, 128, 1, False, True), (200, 17, 9, 6, 1, True, True), (0, 8, 8, 4, 1, True, True),
    (3000, 64, 64, 256, 1, False, False)])
def test_renderer_edge_shapes(hip_lib, N, H, W, K, B, aniso, fused):
    """Shapes at the edges of every fast path of the renderer (no Gaussians at all, a 1x1 image, odd K, K beyond the
    fused backward's 128, K at the library's 256, batches, full 3x3 forms): the frame runs forward and backward, every
    output is finite, an empty scene renders the background, and the shade-through backward is taken exactly where it
    applies (K <= 128; beyond that the image takes voge_shade_bwd and the fragments the one-pass voge_fragment_bwd*)."""
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, to_white_background
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    g = torch.Generator().manual_seed(1)
    verts = torch.rand(N, 3, generator=g) * 2 - 1
    sig = torch.full((N,), 800.0) if not aniso else torch.eye(3)[None].repeat(N, 1, 1) * 800.0
    gm = GaussianMeshes(verts, sig).to(DEV)
    colors = torch.rand(N * B, 3, generator=g).to(DEV).requires_grad_(True)
    R, T = look_at_view_transform(dist=[3.0] * B, elev=[10.0] * B, azim=[30.0 + 20 * b for b in range(B)], device=DEV)
    cams = PerspectiveCameras(focal_length=float(max(H, W)), principal_point=((W / 2.0, H / 2.0),), image_size=((H, W),), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(DEV)
    frag = renderer(gm, R=R, T=T)
    img = to_white_background(frag, colors)
    assert (type(img.grad_fn).__name__ in ("_ShadeThroughBackward", "_CompositeShadeBackward")) == fused
    (img.sum() + get_silhouette(frag).sum()).backward()
    torch.cuda.synchronize()
    assert img.shape == (B, H, W, 3) and torch.isfinite(img).all() and torch.isfinite(colors.grad).all()
    if N:
        assert torch.isfinite(gm.verts.grad).all() and torch.isfinite(gm.sigmas.grad).all()
    else:
        assert (img == 1).all() and (frag.valid_num == 0).all()


@pytest.mark.parametrize("mode", ["stripes", "views", "row_bands"])
def test_bench_two_ranks_on_one_gpu(hip_lib, mode):
    """bench.py's N > 1 path end to end (its own rank spawning, split HIP graphs, overlapped all_gather, flat all_reduce)
    with both ranks on this one GPU through gloo: the line it prints carries the contract's fields for the mode.  The
    default is north_star's split: ONE frame, strong scaling (interleaved stripes)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VOGE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--balance-rounds", "1"]
    if mode != "stripes":
        cmd += ["--no-variants", "--row-bands" if mode == "row_bands" else "--views"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0 and line["unit"] == "frames/s"
    assert line["scaling"] == ("weak" if mode == "views" else "strong") and line["config"]["multi_gpu_mode"] == mode
    # a step is 2 frames (one view per rank) or 1 frame (two ranks' rows)
    frames = 2 if mode == "views" else 1
    assert abs(line["value"] - frames * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert ("bands" in line["config"]) == (mode == "row_bands")
    # the line describes the job it ran on: the process group's world size and backend, every rank's own
    # clock, and the one-off check that the assembled frame is bit for bit the frame one GPU renders
    c = line["config"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["gathered_frame_equals_single_gpu_render"] is True
    assert len(c["rank_ms_per_step"]["per_rank"]) == 2 and 0 < c["rank_ms_per_step"]["min"] <= c["rank_ms_per_step"]["max"] <= line["ms_per_step"] * 1.001
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    if mode == "stripes":      # the default line also carries the alternatives and config 4
        v = line["variants"]
        assert set(v) == {"row_bands_contiguous", "views_weak_scaling", "cfg4_200k_1024_stripes"}
        assert v["views_weak_scaling"]["scaling"] == "weak" and all(x["value"] > 0 for x in v.values())


def test_bench_one_rank_through_rccl(hip_lib):
    """What ONE GPU can check of the N > 1 path under its real backend: VOGE_BENCH_FORCE_DIST=1 makes bench.py take that path with a
    world of one rank -- `init_process_group("nccl", device_id=...)` (RCCL), the split HIP graphs captured with
    capture_error_mode="thread_local" while RCCL's watchdog thread runs, `all_gather_into_tensor` between the two graphs and the
    flat `all_reduce` behind them through RCCL, `all_gather_object` / `barrier` of the timing, and the untimed exactness check
    (the assembled frame == the plain render).  The 2-rank runs of this box go through gloo (above); no hardware with more than
    one GPU has run this path."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, VOGE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    env.pop("VOGE_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-variants",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])      # (the LAST line: bench.py flushes RCCL's stdout banner out first)
    c = line["config"]
    assert c["world_size"] == 1 and c["backend"] == "nccl" and c["multi_gpu_mode"] == "stripes"
    assert c["gathered_frame_equals_single_gpu_render"] is True
    assert "hip graphs (band forward, band backward)" in c["launch"], c["launch"]      # (not the eager fallback)
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["value"] > 0


@pytest.mark.parametrize("form", ["iso", "iso_view_shared", "general", "fragments_iso_lean"])
def test_batch_walked_in_chunks_equals_one_chunk(hip_lib, form):
    """the trace's scratch is sized for a chunk of the batch and the entry points walk the batch in chunks
    (include/voge_hip.h: voge_trace_workspace_bytes).  A three-view batch traced with ONE view's scratch (three chunks: every
    caller-visible array offset per chunk, the chunk's view-local indices moved up afterwards) must equal the same batch
    traced in one chunk, bit for bit -- per-view Gaussian sets, a set shared by all views, full 3x3 forms, and the renderer's
    trace-only form with the records kept."""
    lib = hip_lib
    rng = np.random.default_rng(77)
    B, N, H, W, K = 3, 900, 40, 72, 12
    verts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    sig = rng.uniform(60, 250, (B, N)).astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.2] * B, [12.0] * B, [40.0, 75.0, 130.0])
    rays, origin = camera_np.pixel_rays(R, T, 60.0, (W / 2.0, H / 2.0), (H, W))
    rays_t, org_t = t(rays.astype(np.float32)), t(origin.astype(np.float32))
    cones = torch.empty(int(lib.voge_cones_floats(B, H, W)), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    P_ = lambda x: None if x is None else x.data_ptr()
    assert lib.voge_ray_cones(P_(rays_t), B, H, W, P_(cones), st) == 0
    thr_act = oracle.thr_act_of(0.01)
    mus = t(verts - origin.astype(np.float32)[:, None]).reshape(-1, 3).contiguous()
    a = t(2 * sig).reshape(-1).contiguous()
    isg = (t(2 * sig).reshape(-1, 1, 1) * torch.eye(3, device="cuda")).contiguous()
    isg[::3, 0, 1] = 0.3 * isg[::3, 0, 0]; isg[::3, 1, 0] = isg[::3, 0, 1]            # a third of them anisotropic
    n_full, n_one = lib.voge_trace_workspace_bytes(B, N, H, W), lib.voge_trace_workspace_bytes(1, N, H, W)
    assert n_one < n_full
    outs = []
    for nws in (n_full, n_one):
        ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
        idx = torch.full((B, H, W, K), -7, dtype=torch.int32, device="cuda")
        ln, act, dsd = (torch.full((B, H, W, K), -7.0, device="cuda") for _ in range(3))
        cnt = torch.full((B, H, W), -7, dtype=torch.int32, device="cuda")
        rec = torch.full((B * N, 12 if form == "general" else 4), -7.0, device="cuda")
        if form == "iso":
            rc = lib.voge_trace_topk_fwd_iso(P_(mus), P_(a), P_(rays_t), None, P_(cones), B, N, H, W, K, thr_act, P_(ws), nws,
                                             P_(idx), P_(ln), P_(act), P_(dsd), P_(cnt), st)
        elif form == "iso_view_shared":
            rc = lib.voge_trace_topk_fwd_iso_view(P_(t(verts[0])), P_(t(sig[0])), P_(org_t), 1, 1, P_(rays_t), None, P_(cones), B, N, H, W, K,
                                                  thr_act, P_(ws), nws, P_(idx), P_(ln), P_(act), P_(dsd), P_(cnt), st)
        elif form == "general":
            rc = lib.voge_trace_lean_fwd(P_(mus), P_(isg), P_(rays_t), None, P_(cones), B, N, H, W, K, thr_act, P_(ws), nws,
                                         P_(idx), P_(ln), P_(cnt), P_(rec), st)
        else:
            rc = lib.voge_fragments_fwd_iso(P_(mus), P_(a), P_(rays_t), None, P_(cones), B, N, H, W, K, thr_act, 1.0, P_(ws), nws,
                                            P_(idx), P_(ln), None, None, P_(cnt), None, None, P_(rec), st)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append([n(x) for x in (idx, ln, act, dsd, cnt, rec)])
    one, chunked = outs
    assert (one[0] >= 0).mean() > 0.05 and one[0][2].max() >= 2 * N          # the last view's indices address its own Gaussians
    for name, x, y in zip(("idx", "len", "act", "dsd", "cnt", "records"), one, chunked):
        assert np.array_equal(x, y), (form, name)
