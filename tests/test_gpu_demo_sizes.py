"""The reference's remaining integration settings at their REAL sizes, against the
fp64 oracle on row bands of the frame:
  demo/ReasonOcclusion.py:27-34,52-112      two cuboids (6778 Gaussians), 400x400, max_assign 60, max_point_per_bin 1500,
                                             interpolate_attr + MSE, gradients to the two objects' TRANSLATIONS;
  demo/EfficientCuboidViaOptimization.py:75-79,104-112   102 Gaussians, 256x256, max_assign = 102, thr_activation = 0,
                                             sigmas = L L^T, six-channel attribute, L1 loss;
  demo/ExtractTexture.py:26-47              data/car.off through the reference's loader + converter (25 662 Gaussians,
                                             fixture tests/golden/car_gaussians.npz), 256x672, max_assign 80, focal 1800,
                                             sample_features + re-render from the rotated view.
"""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import camera_np, extras_np
from util import GOLDEN, TOL, _report_flips, grad_close, log_line

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a, dtype=torch.float32, rg=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=rg)


def n(x):
    return x.detach().cpu().numpy()


def oracle_band(verts, sigmas, R, T, focal, pp, size, K, rows, thr=0.01, occ=1.0):
    """The oracle's forward chain up to the weights on pixel rows [rows[0], rows[1]) of the frame."""
    rays, origin = camera_np.pixel_rays(R, T, focal, pp, size)
    rays = np.ascontiguousarray(rays[:, rows[0]:rows[1]])
    mus = (np.asarray(verts, np.float32)[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(np.asarray(sigmas, np.float32))).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(thr)
    idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    w, vn = oracle.composite_fwd(idx, act, ln, dsd, occ)
    return dict(rays=rays, mus=mus, isg=isg, idx=idx, len=ln, act=act, dsd=dsd, weight=w, valid_num=vn, occ=occ)


def same_lists(idx_got, ref, label, max_flips):
    same = (idx_got == np.where(ref["idx"] < 0, 0, ref["idx"])).all(-1) | (idx_got == ref["idx"]).all(-1)
    _report_flips(label, (~same).sum(), same.size)
    assert (~same).sum() <= max_flips, f"{label}: {(~same).sum()} of {same.size} pixels flipped (ceiling {max_flips})"
    return same


def test_reason_occlusion_real_size_translation_gradients(hip_lib):
    from voge_amd.Converter import Cuboid
    from voge_amd.Meshes import GaussianMeshesNaive
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, interpolate_attr
    from voge_amd.cameras import PerspectiveCameras
    pct = 0.7
    c0 = np.array([[0, 0.2, 1], [0, 0.2, 1], [0, 1, 0.2], [0, 1, 0.2], [0, 1, 1], [0, 1, 1]])
    c1 = np.array([[1, 0.2, 0], [1, 0.2, 0], [1, 1, 0], [1, 1, 0], [0.2, 1, 0], [0.2, 1, 0]])
    v0, s0, col0 = Cuboid.cuboid_gauss((-0.8, 0.8), (-0.4, 0.4), (-0.6, 0.6), 4000, colors=c0, percentage=pct)
    v1, s1, col1 = Cuboid.cuboid_gauss((-1, 1), (-1, 1), (-0.3, 0.3), 3000, colors=c1, percentage=pct)
    v0, s0, col0, v1, s1, col1 = (np.asarray(x, np.float32) for x in (v0, s0, col0, v1, s1, col1))
    N0, N1 = len(v0), len(v1)
    assert 6000 < N0 + N1 < 7500, (N0, N1)
    size, K, focal = (400, 400), 60, 300.0
    R, T = camera_np.look_at_view_transform(5.0, 10.0, 20.0)
    cams = PerspectiveCameras(focal_length=focal, principal_point=((200.0, 200.0),), image_size=(size,), device=DEV)
    st = GaussianRenderSettings(max_assign=K, principal=(200, 200), image_size=size, max_point_per_bin=1500)
    renderer = GaussianRenderer(cameras=cams, render_settings=st).to(DEV)
    sig = np.concatenate((s0, s1))
    cols = np.concatenate((col0, col1))
    base0, base1 = np.array([[0.5, 0, 1]], np.float32), np.zeros((1, 3), np.float32)
    # a state half way through the demo's optimisation: the first cuboid 0.9 units off its target, the second 0.3
    p0, p1 = np.array([[-0.1, 0.2, 0.4]], np.float32), np.array([[0.2, -0.1, 0.1]], np.float32)
    with torch.no_grad():
        frag = renderer(GaussianMeshesNaive(t(np.concatenate((v0 + base0, v1 + base1))), t(sig)), R=t(R), T=t(T))
        timg = interpolate_attr(frag, t(cols))
    vp0, vp1 = t(p0, rg=True), t(p1, rg=True)
    verts_t = torch.cat((t(v0) + vp0, t(v1) + vp1), dim=0)
    frag = renderer(GaussianMeshesNaive(verts_t, t(sig)), R=t(R), T=t(T))
    img = interpolate_attr(frag, t(cols))
    rows = (120, 280)
    ref = oracle_band(np.concatenate((v0 + p0, v1 + p1)), sig, R, T, focal, (200.0, 200.0), size, K, rows)
    assert (ref["valid_num"] == K).mean() > 0.03 and (ref["valid_num"] > 0).mean() > 0.25     # full lists and a rim
    same = same_lists(n(frag.vert_index)[:, rows[0]:rows[1]], ref, "ReasonOcclusion 400^2 K=60 (rows 120..279)", max_flips=40)
    rgb_ref = oracle.merge_fwd(cols, ref["idx"], ref["weight"], ref["valid_num"])
    assert np.abs(n(img)[:, rows[0]:rows[1]] - rgb_ref)[same].max() < TOL
    keep = np.zeros((1,) + size, np.float32)
    keep[:, rows[0]:rows[1]] = same
    npx = float(np.prod(size)) * 3
    (((img - timg) ** 2) * t(keep)[..., None]).sum().div(npx).backward()                   # MSELoss on the band's agreed pixels
    g_rgb = 2 * (rgb_ref - n(timg)[:, rows[0]:rows[1]].astype(np.float64)) / npx * same[..., None]
    _, g_w = oracle.merge_bwd(cols, ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w, 1.0)
    _, g_mu, _ = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
    g_mu = g_mu.reshape(-1, 3)
    want0, want1 = g_mu[:N0].sum(0), g_mu[N0:].sum(0)
    for name, got, want in (("v_pred0", vp0.grad, want0), ("v_pred1", vp1.grad, want1)):
        err = np.abs(n(got).reshape(3).astype(np.float64) - want).max()
        log_line(f"[parity] ReasonOcclusion {name}: max err {err:.3e}, largest entry {np.abs(want).max():.3e}")
        assert np.abs(want).max() > 0 and err <= 2 * TOL * np.abs(want).max(), (name, err, want)


def test_efficient_cuboid_stage_two_real_size(hip_lib):
    from voge_amd.Meshes import GaussianMeshesNaive
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, interpolate_attr
    from voge_amd.cameras import PerspectiveCameras
    N, size, focal = 102, (256, 256), 200.0
    K = N
    rng = np.random.default_rng(23)
    face = np.repeat(np.arange(6), 17)
    verts = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    verts[np.arange(N), face % 3] = np.where(face < 3, 1.0, -1.0)
    L0 = (np.eye(3)[None] * 2.0 + np.tril(rng.uniform(-0.3, 0.3, (N, 3, 3)))).astype(np.float32)      # sig_ori around 4 I
    L = t(L0, rg=True)
    sig_t = L @ L.transpose(1, 2)                                                                    # to_sym of the demo
    attr = np.eye(6, dtype=np.float32)[face]
    R, T = camera_np.look_at_view_transform(5.0, -35.0, 200.0)
    cams = PerspectiveCameras(focal_length=focal, principal_point=((128.0, 128.0),), image_size=(size,), device=DEV)
    st = GaussianRenderSettings(max_assign=K, principal=(128, 128), image_size=size, max_point_per_bin=-1, thr_activation=0)
    renderer = GaussianRenderer(cameras=cams, render_settings=st).to(DEV)
    verts_t = t(verts, rg=True)
    frag = renderer(GaussianMeshesNaive(verts_t, sig_t), R=t(R), T=t(T))
    img = interpolate_attr(frag, t(attr))
    rows = (64, 192)
    ref = oracle_band(verts, n(sig_t), R, T, focal, (128.0, 128.0), size, K, rows, thr=0.0)
    assert ref["valid_num"].max() > 40
    same = same_lists(n(frag.vert_index)[:, rows[0]:rows[1]], ref, "EfficientCuboid stage 2, 256^2 K=102 thr 0 (rows 64..191)", max_flips=60)
    rgb_ref = oracle.merge_fwd(attr, ref["idx"], ref["weight"], ref["valid_num"])
    assert np.abs(n(img)[:, rows[0]:rows[1]] - rgb_ref)[same].max() < TOL
    tgt = rng.uniform(0, 1, rgb_ref.shape)
    keep = np.zeros((1,) + size, np.float32)
    keep[:, rows[0]:rows[1]] = same
    tgt_full = np.zeros((1,) + size + (6,))
    tgt_full[:, rows[0]:rows[1]] = tgt
    ((img - t(tgt_full)).abs() * t(keep)[..., None]).sum().backward()                                 # L1Loss (sum-reduced)
    g_rgb = np.sign(rgb_ref - tgt) * same[..., None]
    _, g_w = oracle.merge_bwd(attr, ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w, 1.0)
    _, g_mu, g_A = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
    g_sig = 2 * g_A.reshape(N, 3, 3)
    grad_close("EfficientCuboid 256^2 verts", n(verts_t.grad), g_mu.reshape(N, 3), 0.5 * TOL)
    g_L = (g_sig + g_sig.transpose(0, 2, 1)) @ L0.astype(np.float64)
    grad_close("EfficientCuboid 256^2 L", n(L.grad), g_L, 0.5 * TOL)


def test_extract_texture_real_size_car(hip_lib):
    from voge_amd.Meshes import GaussianMeshesNaive
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.Sampler import sample_features
    from voge_amd.Utils import rotation_theta
    from voge_amd.cameras import PerspectiveCameras
    g = np.load(os.path.join(GOLDEN, "car_gaussians.npz"))
    verts, sig = g["verts"], g["isigma"]
    N = verts.shape[0]
    assert N == 25662 and sig.shape == (N,)
    size, K, focal, pp = (256, 672), 80, 1800.0, (336.0, 128.0)
    st = GaussianRenderSettings(batch_size=-1, image_size=size, max_assign=K)                       # (max_point_per_bin=None: the demo's default)
    cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=(size,), device=DEV)
    render = GaussianRenderer(cameras=cams, render_settings=st).to(DEV)
    theta, azim, elev = float(g["theta"]), float(g["azimuth"]), float(g["elevation"])
    rows = (96, 160)
    rng = np.random.default_rng(3)
    photo = rng.uniform(0, 255, (1,) + size + (3,)).astype(np.float32)
    textures, refs = [], []
    for a in (azim, azim - math.pi / 6):                                                            # the sampled view, the rotated one
        R0, T0 = camera_np.look_at_view_transform(3.0, math.degrees(elev), math.degrees(a))
        Rz = n(rotation_theta(torch.tensor([theta])))[0].astype(np.float64)
        R = (R0[0] @ Rz)[None].astype(np.float32)
        with torch.no_grad():
            frag = render(GaussianMeshesNaive(t(verts), t(sig)), R=t(R), T=t(T0), rows=rows)
        ref = oracle_band(verts, sig, R, T0, focal, pp, size, K, rows)
        same = same_lists(n(frag.vert_index), ref, f"ExtractTexture car 256x672 K=80, azimuth {a:+.2f} (rows 96..159)", max_flips=120)
        assert (ref["valid_num"] > 0).mean() > 0.2
        assert np.abs(n(frag.vert_weight) - ref["weight"])[same].max() < TOL
        refs.append((frag, ref, same))
    # ---- sample_features on the first view's band (ExtractTexture.py:47-49) against the restated sampler
    frag, ref, same = refs[0]
    band = photo[:, rows[0]:rows[1]]
    with torch.no_grad():
        get, get_sum = sample_features(frag, t(band), N)
    rf, rw = extras_np.sample_voge(band, ref["weight"] * same[..., None], np.where(same[..., None], ref["idx"], -1), N)
    mask = t(same.astype(np.float32))
    with torch.no_grad():      # (the same comparison restricted to the pixels whose lists agree)
        fr2 = type(frag)(vert_weight=frag.vert_weight * mask[..., None], vert_index=frag.vert_index, valid_num=frag.valid_num,
                         vert_hit_length=frag.vert_hit_length)
        get, get_sum = sample_features(fr2, t(band), N)
    assert np.abs(n(get_sum) - rw).max() <= TOL * max(1.0, np.abs(rw).max())
    assert np.abs(n(get) - rf).max() <= TOL * max(1.0, np.abs(rf).max())
    texture = (get / (1e-8 + get_sum[:, None]) / 255 * 0.7)
    # ---- the rotated view re-rendered with the extracted texture (:51-57)
    frag2, ref2, same2 = refs[1]
    with torch.no_grad():
        img = to_white_background(frag2, texture)
    tex64 = n(texture).astype(np.float64)
    rgb = oracle.merge_fwd(tex64, ref2["idx"], ref2["weight"], ref2["valid_num"])
    want, _ = oracle.blend_fwd(rgb, ref2["weight"])
    assert np.abs(n(img) - want)[same2].max() < TOL
    assert float((n(img) < 0.99).mean()) > 0.05
