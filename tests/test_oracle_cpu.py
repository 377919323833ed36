"""CPU tests: the oracle against the golden fixtures produced by the imported reference
(tests/golden/make_golden.py) and against the reference's embedded known-answer vector."""
import os

import numpy as np
import pytest

import oracle
from oracle import camera_np
from util import GOLDEN


@pytest.mark.parametrize("name", ["k5", "k25", "k40"])
def test_composite_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, f"composite_{name}.npz"))
    w, vn = oracle.composite_fwd(g["idx"], g["act"], g["len"], g["dsd"], float(g["occ"]))
    assert np.abs(w - g["weight"]).max() < 1e-14
    assert (vn == g["valid_num"]).all()
    # empty slots give exactly zero weight
    assert (w[g["idx"] < 0] == 0).all()
    ga, gl, gd = oracle.composite_bwd(g["act"], g["len"], g["dsd"], g["g_weight"], float(g["occ"]))
    for got, key in ((ga, "g_act"), (gl, "g_len"), (gd, "g_dsd")):
        ref = g[key]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), key
    # the reference's own fp32 run stays within the stated tolerance of the fp64 truth
    assert np.abs(g["weight_f32"] - g["weight"]).max() < 1e-4


def test_merge_blend_matches_reference():
    m = np.load(os.path.join(GOLDEN, "merge_blend.npz"))
    rgb = oracle.merge_fwd(m["colors"], m["idx"], m["weight"], m["valid_num"])
    assert np.abs(rgb - m["rgb"]).max() < 1e-14
    fm = oracle.merge_fwd(m["feat"], m["idx"], m["weight"], m["valid_num"])
    assert np.abs(fm - m["feat_map"]).max() < 1e-13
    img, sil = oracle.blend_fwd(rgb, m["weight"])
    assert np.abs(img - m["img_white"]).max() < 1e-14
    assert np.abs(sil - m["silhouette"]).max() < 1e-14
    img2, _ = oracle.blend_fwd(rgb, m["weight"], m["bg"], float(m["thr"]))
    assert np.abs(img2 - m["img_colored_thr"]).max() < 1e-6   # reference keeps bg in fp32
    # in-place index fix of merge_final: -1 -> 0
    assert (m["idx_after_merge"] == np.where(m["idx"] < 0, m["idx"] + 1, m["idx"])).all()
    g_attr, g_w = oracle.merge_bwd(m["feat"], m["idx"], m["weight"], m["valid_num"], m["g_feat"])
    assert np.abs(g_attr - m["g_feat_attr"]).max() < 1e-12


def test_trace_backward_known_answer():
    """ray_trace_voge.cu:381-448: grad_mu=[1,.78,0], grad_isigma=[[-.04,.24,0],[-.18,.04,0],0], grad_ray=[-1.3,-.944,0]."""
    k = np.load(os.path.join(GOLDEN, "trace_bwd_known_answer.npz"))
    ray = k["ray"].reshape(1, 1, 1, 3)
    idx, ln, act, dsd = oracle.trace_fwd(k["mu"][None], k["isigma"][None], ray, 1, 1e9)
    assert idx.item() == 0
    assert abs(ln.item() - float(k["len"])) < 1e-6 and abs(act.item() - float(k["act"])) < 1e-6
    one, zero = np.ones((1, 1, 1, 1)), np.zeros((1, 1, 1, 1))
    g_ray, g_mu, g_isg = oracle.trace_bwd(k["mu"][None], k["isigma"][None], ray, idx, one, one, zero)
    assert np.abs(g_mu[0] - [1.0, 0.78, 0.0]).max() < 1e-6
    assert np.abs(g_isg[0] - [[-0.04, 0.24, 0], [-0.18, 0.04, 0], [0, 0, 0]]).max() < 1e-6
    assert np.abs(g_ray.reshape(3) - [-1.3, -0.944, 0.0]).max() < 1e-6
    assert np.abs(g_mu[0] - k["g_mu"]).max() < 1e-6 and np.abs(g_isg[0] - k["g_isigma"]).max() < 1e-6


def _dense_trace(mus, isg, rays, K, thr_act):
    """Independent numpy formulation: dense quadratic forms + stable argsort."""
    B, H, W, _ = rays.shape
    d = rays.reshape(B, -1, 3).astype(np.float64)
    mu = mus.reshape(B, -1, 3).astype(np.float64)
    A = isg.reshape(B, -1, 3, 3).astype(np.float64)
    N = mu.shape[1]
    out = []
    for b in range(B):
        Ad = np.einsum("nij,pj->pni", A[b], d[b])
        ksk = np.einsum("pi,pni->pn", d[b], Ad)
        msk = np.einsum("ni,pni->pn", mu[b], Ad)
        msm = np.einsum("ni,nij,nj->n", mu[b], A[b], mu[b])[None]
        ln = msk / ksk
        act = msm - msk * msk / ksk
        key = np.where(act < thr_act, ln, np.inf)
        order = np.argsort(key, axis=1, kind="stable")[:, :K]
        ok = np.take_along_axis(key, order, 1) < 1e10
        out.append((np.where(ok, order + b * N, -1), np.where(ok, np.take_along_axis(ln, order, 1), 1e10),
                    np.where(ok, np.take_along_axis(act, order, 1), 1e10),
                    np.where(ok, np.take_along_axis(ksk, order, 1), 0.0)))
    return [np.stack([o[i] for o in out]).reshape(B, H, W, K) for i in range(4)]


def test_trace_forward_vs_dense_formulation():
    rng = np.random.default_rng(3)
    B, N, H, W, K = 2, 150, 9, 13, 6
    mus = rng.normal(size=(B, N, 3)).astype(np.float32) + np.float32([0, 0, 4])
    L = np.tril(rng.uniform(-1, 1, (B, N, 3, 3))) * 3
    isg = (L @ L.transpose(0, 1, 3, 2) + 0.5 * np.eye(3)).astype(np.float32)
    rays = rng.normal(size=(B, H, W, 3)) * 0.3 + [0, 0, 1]
    rays = (rays / np.linalg.norm(rays, axis=-1, keepdims=True)).astype(np.float32)
    thr = oracle.thr_act_of(0.01)
    got = oracle.trace_fwd(mus, isg, rays, K, thr)
    ref = _dense_trace(mus, isg, rays, K, thr)
    assert (got[0] == ref[0]).all()
    for g, r in zip(got[1:], ref[1:]):
        assert np.abs(g - r).max() < 1e-9
    # explicit bin lists: all-candidate list in every bin == NULL list (RayTracing.py:22-26)
    bs = 8
    BH, BW = (H - 1) // bs + 1, (W - 1) // bs + 1
    bins = (np.arange(N)[None, None, None, :] + np.arange(B)[:, None, None, None] * N) * np.ones((1, BH, BW, 1), int)
    got2 = oracle.trace_fwd(mus, isg, rays, K, thr, bin_points=bins.astype(np.int32), bin_size=bs)
    for a, b in zip(got, got2):
        assert (a == b).all()
    # K larger than the number of hits: tail is sentinels
    got3 = oracle.trace_fwd(mus, isg, rays, 200, thr)
    assert (got3[0][..., -1] == -1).all() and (got3[1][..., -1] == 1e10).all()


def test_trace_backward_vs_finite_differences():
    rng = np.random.default_rng(5)
    N, H, W, K = 7, 2, 3, 3
    mus = (rng.normal(size=(N, 3)) * 0.2 + [0, 0, 3]).astype(np.float32)
    L = np.tril(rng.uniform(0.5, 1.5, (N, 3, 3)))
    isg = (L @ L.transpose(0, 2, 1) + rng.normal(size=(N, 3, 3)) * 0.05).astype(np.float32)  # not symmetric
    rays = rng.normal(size=(1, H, W, 3)) * 0.05 + [0, 0, 1]
    rays = (rays / np.linalg.norm(rays, axis=-1, keepdims=True)).astype(np.float32)
    idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, K, 1e9)
    gl, ga, gd = (rng.normal(size=idx.shape) for _ in range(3))
    g_ray, g_mu, g_isg = oracle.trace_bwd(mus, isg, rays, idx, gl, ga, gd)

    def loss(m, a, r):
        tot = 0.0
        for (y, x, k), p in np.ndenumerate(idx[0]):
            d = r[0, y, x].astype(np.float64)
            A, mu = a[p].astype(np.float64), m[p].astype(np.float64)
            ksk, msk, msm = d @ A @ d, mu @ A @ d, mu @ A @ mu
            tot += gl[0, y, x, k] * msk / ksk + ga[0, y, x, k] * (msm - msk * msk / ksk) + gd[0, y, x, k] * ksk
        return tot
    eps = 1e-3  # inputs are fp32: perturb on a grid exactly representable around the values
    for arr, grad, pos in ((mus, g_mu, (2, 1)), (isg, g_isg, (3, 0, 2)), (isg, g_isg, (3, 2, 0)), (rays, g_ray, (0, 1, 2, 0))):
        hi, lo = arr.astype(np.float64), arr.astype(np.float64)
        hi[pos] += eps
        lo[pos] -= eps
        args_hi = [mus, isg, rays]
        args_lo = [mus, isg, rays]
        i = 0 if arr is mus else (1 if arr is isg else 2)
        args_hi[i], args_lo[i] = hi, lo
        fd = (loss(*args_hi) - loss(*args_lo)) / (2 * eps)
        assert abs(fd - grad[pos]) <= 1e-5 * max(1.0, abs(fd)), (pos, fd, grad[pos])


def test_camera_conventions():
    R, T = camera_np.look_at_view_transform(6, 10, 70)
    rays, C = camera_np.pixel_rays(R, T, 300.0, (128.0, 128.0), (256, 256))
    assert np.abs(np.linalg.norm(rays, axis=-1) - 1).max() < 1e-6
    assert np.abs(C[0] - 6 * np.array([np.cos(np.deg2rad(10)) * np.sin(np.deg2rad(70)), np.sin(np.deg2rad(10)),
                                       np.cos(np.deg2rad(10)) * np.cos(np.deg2rad(70))])).max() < 1e-6
    # a world point projects (x_view = px - fx X/Z ...) onto the pixel whose ray passes through it
    P = np.array([0.3, -0.2, 0.1])
    v = P @ R[0].astype(np.float64) + T[0]
    col, row = 128 - 300 * v[0] / v[2], 128 - 300 * v[1] / v[2]
    i, j = int(np.floor(row)), int(np.floor(col))
    d = (P - C[0]) / np.linalg.norm(P - C[0])
    assert np.arccos(np.clip(rays[0, i, j] @ d, -1, 1)) < 1.5 / 300


def test_expend_sigma_and_whole_frame():
    g = np.load(os.path.join(GOLDEN, "misc_api.npz"))
    assert (camera_np.expand_sigma(np.float32([1.5, 2.5])) == g["expend_1"]).all()
    assert (camera_np.expand_sigma(np.float32([[1, 2, 3], [4, 5, 6]])) == g["expend_2"]).all()
    R, T = camera_np.look_at_view_transform(6, 10, 70)
    verts, isig = g["cuboid_verts"], g["cuboid_isigma"]
    out = oracle.render(verts, isig, (verts + 1) / 3, R, T, 60.0, (32.0, 32.0), (64, 64), K=20)
    assert out["image"].shape == (1, 64, 64, 3) and np.isfinite(out["image"]).all()
    assert out["weight"].min() >= 0 and out["image"].max() <= 1.0
    assert out["valid_num"].max() > 3 and (out["image"][0, 0, 0] == 1).all()  # corner is background
    valid = out["idx"] >= 0
    assert (np.diff(np.where(valid, out["len"], np.inf), axis=-1)[valid[..., 1:]] >= 0).all()


def test_next_row_restatements_are_self_consistent():
    """oracle/extras_np.py: dense backward vs central differences, nearest-K vs a stable sort,
    sampler backward vs the bilinear identity <g, sample(image)> = <image, g_img>."""
    from oracle import extras_np
    rng = np.random.default_rng(9)
    M, N = 5, 7
    mus = rng.normal(size=(M, 3)) * 0.3 + [0, 0, 3]
    L = np.tril(rng.uniform(0.5, 1.5, (M, 3, 3)))
    A = L @ L.transpose(0, 2, 1) + rng.normal(size=(M, 3, 3)) * 0.05
    d = rng.normal(size=(N, 3)) * 0.1 + [0, 0, 1]
    gl, ga, gd = (rng.normal(size=(N, M)) for _ in range(3))
    loss = lambda m, a, r: sum((x * g).sum() for x, g in zip(extras_np.ray_dense_fwd(m, a, r), (gl, ga, gd)))
    g_ray, g_mu, g_A = extras_np.ray_dense_bwd(mus, A, d, gl, ga, gd)
    eps = 1e-6
    for arr, grad, pos in ((mus, g_mu, (2, 1)), (A, g_A, (3, 0, 2)), (A, g_A, (3, 2, 0)), (d, g_ray, (4, 1))):
        hi, lo = arr.copy(), arr.copy()
        hi[pos] += eps
        lo[pos] -= eps
        args = [mus, A, d]
        i = 0 if arr is mus else (1 if arr is A else 2)
        fd = (loss(*[hi if j == i else x for j, x in enumerate(args)]) - loss(*[lo if j == i else x for j, x in enumerate(args)])) / (2 * eps)
        assert abs(fd - grad[pos]) <= 1e-6 * max(1.0, abs(fd))
    ln, act, dsd = extras_np.ray_dense_fwd(mus, A, d)
    idx, ol, oa, od = extras_np.find_nearest_k(ln, act, dsd, 3, np.median(act))
    for r in range(N):
        order = [m for m in np.argsort(ln[r], kind="stable") if act[r, m] < np.median(act)][:3]
        assert idx[r, :len(order)].tolist() == order and (idx[r, len(order):] == -1).all()
    image, w = rng.uniform(size=(1, 3, 4, 2)), rng.uniform(size=(1, 3, 4, 5))
    ix = rng.integers(-1, 6, (1, 3, 4, 5)).astype(np.int32)
    feat, wsum = extras_np.sample_voge(image, w, ix, 6)
    gF, gW = rng.normal(size=feat.shape), rng.normal(size=wsum.shape)
    g_img, g_w = extras_np.sample_voge_bwd(image, w, ix, gF, gW)
    assert abs((feat * gF).sum() - (image * g_img).sum()) < 1e-10          # linear in image
    assert abs((feat * gF).sum() + (wsum * gW).sum() - (w * g_w).sum()) < 1e-10  # and in w


def test_coarse_stage_restatement():
    """oracle/coarse_np.py (RayTracing.py:33-73, rasterize_coarse.cu:20-188): the bin test against a direct per-bin
    evaluation, the half-pixel pad, the z < 0 skip, ascending lists, and the chunk-drop overflow rule."""
    from oracle import coarse_np
    rng = np.random.default_rng(0)
    P, H, W, bs = 1300, 50, 70, 10
    pts = np.stack([rng.uniform(-1.6, 1.6, P), rng.uniform(-1.2, 1.2, P), rng.uniform(-0.5, 4, P)], 1).astype(np.float32)
    rad = rng.uniform(0.01, 0.3, (P, 2)).astype(np.float32)
    rad[5] = np.nan                                                     # negative column sum in convert_to_box
    first, num = np.array([0, 700], np.int64), np.array([700, 600], np.int64)
    bins = coarse_np.rasterize_points_coarse(pts, first, num, (H, W), rad, bs, 2000)
    assert bins.shape == (2, 5, 7, 2000)
    # direct evaluation of one bin: NDC x range is 2 * W / H wide (W > H), pixel centres at half-pixel offsets
    by, bx = 2, 4
    xr, yr = 2.0 * W / H, 2.0
    x0, x1 = -xr / 2 + xr * (bx * bs) / W, -xr / 2 + xr * ((bx + 1) * bs) / W
    y0, y1 = -yr / 2 + yr * (by * bs) / H, -yr / 2 + yr * ((by + 1) * bs) / H
    e = np.arange(700)
    with np.errstate(invalid="ignore"):
        hit = ((pts[e, 0] - rad[e, 0] <= x1 + 1e-6) & (x0 - 1e-6 < pts[e, 0] + rad[e, 0]) & (pts[e, 1] - rad[e, 1] <= y1 + 1e-6)
               & (y0 - 1e-6 < pts[e, 1] + rad[e, 1]) & ~(pts[e, 2] < 0))
    got = bins[0, by, bx]
    got = got[got >= 0]
    assert np.array_equal(got, np.sort(got)) and set(got.tolist()) <= set(e[hit].tolist())
    assert len(set(e[hit].tolist()) - set(got.tolist())) <= 2          # (only the 1e-6 slack of this check)
    assert 5 not in bins[0] and (bins[1][bins[1] >= 0] >= 700).all()   # NaN radius never listed; batch 1 lists its own points
    assert not np.isin(np.nonzero(pts[:, 2] < 0)[0], bins).any()
    # overflow: a bin holds M = 40 entries; the chunk (512 points) that does not fit is dropped entirely
    small = coarse_np.rasterize_points_coarse(pts, first, num, (H, W), rad, bs, 40)
    full_cnt, small_cnt = (bins[0] >= 0).sum(-1), (small[0] >= 0).sum(-1)
    assert (small_cnt <= full_cnt).all() and (small_cnt < full_cnt).any()
    chunk0 = (bins[0] >= 0) & (bins[0] < 512)
    fits = chunk0.sum(-1) <= 40
    assert ((small_cnt >= chunk0.sum(-1)) | ~fits).all()               # the first chunk is kept whenever it fits
    # projection: an isotropic Gaussian straight ahead projects to the principal point with radius sqrt(-ln thr / a) f 2/s / Z
    R, T = np.eye(3)[None], np.array([[0.0, 0.0, 4.0]])
    p, r = coarse_np.project_for_coarse(np.array([[[0.0, 0.0, 4.0], [0.4, 0.0, 4.0]]]), np.eye(3)[None, None] * np.array([50.0, 50.0])[None, :, None, None],
                                        R, T, 100.0, (35.0, 25.0), (50, 70), 0.01)
    assert np.allclose(p[0, 0], [(35 - 35) * 2 / 50, (25 - 25) * 2 / 50, 4.0])
    assert np.allclose(p[0, 1, 0], (35 - 100 * 0.4 / 4 - 35) * 2 / 50)              # +X is left: the column decreases
    assert np.allclose(r[0, 0], np.sqrt(-np.log(0.01) / 50.0) * 100 * 2 / 50 / 4.0, rtol=1e-5)


def test_ray_convention_against_the_reference_get_ray_camera_space():
    """VoGE/Aggregation.py:11-27 (`get_ray_camera_space`), run by make_golden.py: the reference's own statement of the
    view-space ray directions -- x = -(col - px) / fx, y = -(row - py) / fy, z = 1, normalised, sampled at pixel
    CORNERS, `principle` ordered (row, col).  oracle/camera_np.pixel_rays (PyTorch3D's convention: pixel CENTRES)
    must give the same directions when the principal point is moved by half a pixel; with R = I, T = 0 world space is
    view space."""
    g = np.load(os.path.join(GOLDEN, "ray_camera_space.npz"))
    for name in "abc":
        H, W = (int(v) for v in g[name + "_size"])
        py, px = g[name + "_principle_row_col"]
        fy, fx = g[name + "_focal_row_col"]
        rays, origin = camera_np.pixel_rays(np.eye(3)[None], np.zeros((1, 3)), (fx, fy), (px + 0.5, py + 0.5), (H, W))
        assert np.abs(origin).max() == 0
        assert np.abs(rays[0] - g[name + "_dirs"]).max() < 2e-7, name
        # and WITHOUT the shift the two differ by exactly the half-pixel offset (so the test above is not vacuous)
        raw, _ = camera_np.pixel_rays(np.eye(3)[None], np.zeros((1, 3)), (fx, fy), (px, py), (H, W))
        assert np.abs(raw[0] - g[name + "_dirs"]).max() > 0.2 / max(fx, fy) / 2


# ------------------------------------------------------------------ the torch tensor program of bench.py's cpu_baseline_torch
@pytest.mark.parametrize("name", ["k5", "k25", "k40"])
def test_torch_tensor_program_matches_the_reference_outputs(name):
    """oracle/torch_ref.py restates the reference's dense tensor program (Aggregation.py:30-141, Renderer.py:157-171) for
    the CPU baseline BASELINE.md section 3 defines; its weights and autograd gradients equal what the imported reference
    produced on the same inputs (the golden fixtures of tests/golden/make_golden.py)."""
    import torch
    from oracle import torch_ref
    g = np.load(os.path.join(GOLDEN, f"composite_{name}.npz"))
    td = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=True)
    act, ln, dsd = td(g["act"]), td(g["len"]), td(g["dsd"])
    w, vn = torch_ref.aggregation(torch.tensor(g["idx"]), act, ln, dsd, float(g["occ"]))
    assert np.abs(w.detach().numpy() - g["weight"]).max() < 1e-13 and (vn.numpy() == g["valid_num"]).all()
    (w * torch.tensor(g["g_weight"])).sum().backward()
    for got, key in ((act.grad, "g_act"), (ln.grad, "g_len"), (dsd.grad, "g_dsd")):
        assert np.abs(got.numpy() - g[key]).max() <= 1e-11 * max(1.0, np.abs(g[key]).max()), key


def test_torch_tensor_program_merge_blend_and_dense_trace():
    import torch
    from oracle import torch_ref
    m = np.load(os.path.join(GOLDEN, "merge_blend.npz"))
    td = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    rgb = torch_ref.merge_final(td(m["colors"]), td(m["weight"]), torch.tensor(m["valid_num"]), torch.tensor(m["idx"]))
    assert np.abs(rgb.numpy() - m["rgb"]).max() < 1e-14
    img = torch_ref.to_colored_background(rgb, td(m["weight"]), torch.ones(3, dtype=torch.float64))
    assert np.abs(img.numpy() - m["img_white"]).max() < 1e-14
    img2 = torch_ref.to_colored_background(rgb, td(m["weight"]), td(m["bg"]), float(m["thr"]))
    assert np.abs(img2.numpy() - m["img_colored_thr"]).max() < 1e-6
    # the dense trace against the C oracle's (same selection rule: K smallest len among act < thr_act)
    from util import random_scene
    verts, sig, _ = random_scene(300, seed=3, lo=0.08, hi=0.2)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 25.0)
    H, W, K = 12, 16, 9
    rays, origin = camera_np.pixel_rays(R, T, 20.0, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(0.01)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    idx, ln, act, dsd = torch_ref.trace_dense(td(mus[0]), td(isg[0]), td(rays.reshape(-1, 3)), K, thr_act, chunk=50)
    same = (idx.numpy().reshape(ref[0].shape) == ref[0]).all(-1)
    assert same.mean() > 0.99
    for got, want in ((ln, ref[1]), (act, ref[2]), (dsd, ref[3])):
        assert np.abs(got.numpy().reshape(want.shape)[same] - want[same]).max() < 1e-6 * max(1.0, np.abs(want[same]).max())
