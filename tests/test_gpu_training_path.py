"""GPU parity of the call pattern the reference's own training loops use: they differentiate through
interpolate_attr (merge_final) and get_silhouette -- never through to_white_background
(demo/ShapeFitting.py:217,262-271,295; demo/ReasonOcclusion.py:106-109; demo/EfficientCuboidViaOptimization.py:109-112).
The gradient of the weights then reaches _Fragments.backward, which runs the composite's and the trace's backward as
ONE pass (voge_fragment_bwd_iso / voge_fragment_bwd) for any K <= 256, odd K included.

Everything here is checked against the fp64 oracle chain (composite_bwd -> trace_bwd), not against other kernels of
this repo.
"""
import gc
import weakref

import numpy as np
import pytest
import torch

import oracle
from oracle import camera_np
from util import TOL, grad_close, random_scene, _report_flips

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a, dtype=torch.float32, rg=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=rg)


def n(x):
    return x.detach().cpu().numpy()


def oracle_frame(verts, sigmas, R, T, focal, pp, size, K, thr=0.01, occ=1.0, inverse=False):
    """Forward chain of the oracle up to the weights.  sigmas [N] | [N,3] | [N,3,3] as the user passes them."""
    rays, origin = camera_np.pixel_rays(R, T, focal, pp, size)
    B = rays.shape[0]
    mus = (np.asarray(verts, np.float32)[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    sig3 = camera_np.expand_sigma(np.asarray(sigmas, np.float32))
    isg = (2 * np.linalg.inv(sig3.astype(np.float64))).astype(np.float32) if inverse else (2 * sig3).astype(np.float32)
    isg = np.ascontiguousarray(np.broadcast_to(isg[None], (B,) + isg.shape))
    thr_act = oracle.thr_act_of(thr)
    idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    w, vn = oracle.composite_fwd(idx, act, ln, dsd, occ)
    return dict(rays=rays, mus=mus, isg=isg, idx=idx, len=ln, act=act, dsd=dsd, weight=w, valid_num=vn, occ=occ)


def oracle_param_grads(ref, sigmas, g_w, g_hitlen=None):
    """(g_weight [, g_hit_length]) -> gradients of the user's verts [N,3] and sigmas (their own shape; 2*sigma rule of
    Renderer.py:133: d/d sigma of A = 2 expand(sigma)), summed over the views that share the Gaussians."""
    g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w, ref["occ"])
    if g_hitlen is not None:
        g_len = g_len + g_hitlen
    _, g_mu, g_A = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
    B = ref["rays"].shape[0]
    N = ref["mus"].shape[1]
    g_mu = g_mu.reshape(B, N, 3).sum(0)
    g_A = g_A.reshape(B, N, 3, 3).sum(0)
    sigmas = np.asarray(sigmas)
    if sigmas.ndim == 1:
        g_sig = 2 * np.einsum("nii->n", g_A)
    elif sigmas.ndim == 2:
        g_sig = 2 * np.einsum("nii->ni", g_A)
    else:
        g_sig = 2 * g_A
    return g_mu, g_sig


def same_lists(frag, ref, label, max_flips):
    idx = n(frag.vert_index)
    same = (idx == np.where(ref["idx"] < 0, 0, ref["idx"])).all(-1) | (idx == ref["idx"]).all(-1)
    _report_flips(label, (~same).sum(), same.size)
    assert (~same).sum() <= max_flips, f"{label}: {(~same).sum()} of {same.size} pixels flipped (ceiling {max_flips})"
    assert np.abs(n(frag.vert_weight)[same] - ref["weight"][same]).max(initial=0.0) < TOL
    return same


def renderer_for(H, W, K, focal, thr=0.01, occ=1.0, inverse=False, mppb=-1):
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings
    from voge_amd.cameras import PerspectiveCameras
    cams = PerspectiveCameras(focal_length=focal, principal_point=((W / 2.0, H / 2.0),), image_size=((H, W),), device=DEV)
    st = GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=thr, absorptivity=occ, inverse_sigma=inverse,
                                max_point_per_bin=mppb)
    return GaussianRenderer(cams, st).to(DEV)


def forbid_three_kernel_chain(monkeypatch):
    """Make the stand-alone backward kernels unreachable: a test that passes took the one-pass kernel."""
    from voge_amd import _lib
    lib = _lib.load()

    def boom(*a):
        raise AssertionError("the three-kernel backward chain was taken")
    for name in ("voge_composite_bwd", "voge_trace_bwd", "voge_trace_bwd_iso", "voge_trace_bwd_iso_view", "voge_fragment_act_dsd_iso"):
        monkeypatch.setattr(lib, name, boom, raising=True)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,B,form,inverse", [
    (25, 1, "scalar", False), (40, 2, "scalar", False), (7, 1, "scalar", True), (1, 1, "scalar", False),
    (130, 1, "scalar", False), (131, 1, "scalar", False), (256, 1, "scalar", False),
    (25, 1, "full", False), (12, 2, "full", False), (102, 1, "full", False), (133, 1, "full", False),
    (9, 1, "diag", False), (20, 2, "diag", False)])
def test_fragment_backward_from_a_weight_gradient_vs_oracle(hip_lib, monkeypatch, K, B, form, inverse):
    """Random gradients on vert_weight AND vert_hit_length -> verts / sigmas, against the oracle chain: scalar sigmas
    (voge_fragment_bwd_iso, fragments without act / dsd), full [N,3,3] forms and (N,3) diagonals (voge_fragment_bwd);
    odd K, K > 128 (four slots per lane), batches of views that share the Gaussians."""
    from voge_amd.Meshes import GaussianMeshes
    forbid_three_kernel_chain(monkeypatch)
    N, H, W = (900, 40, 56) if K > 64 else (2000, 56, 72)
    lo, hi = (0.15, 0.3) if K > 64 else (0.05, 0.12)
    verts, sig, _ = random_scene(N, seed=300 + K, lo=lo, hi=hi, aniso=(form == "full"))
    if form == "diag":
        rng = np.random.default_rng(K)
        sig = (sig[:, None] * rng.uniform(0.6, 1.6, (N, 3))).astype(np.float32)
    if inverse:
        sig = (1.0 / sig).astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.0, 3.3][:B], [10.0, -20.0][:B], [30.0, 200.0][:B])
    occ = 1.1
    renderer = renderer_for(H, W, K, 80.0, occ=occ, inverse=inverse)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    frag = renderer(gm, R=t(R), T=t(T))
    ref = oracle_frame(verts, sig, R, T, 80.0, (W / 2.0, H / 2.0), (H, W), K, occ=occ, inverse=inverse)
    same = same_lists(frag, ref, f"weight-gradient form K={K} B={B} {form}", max_flips=12)
    assert (ref["valid_num"] == K).any() or K > 100
    rng = np.random.default_rng(7)
    live = (np.arange(K)[None, None, None] < ref["valid_num"][..., None]) & same[..., None]
    g_w = rng.normal(size=ref["weight"].shape) * live
    g_h = 0.05 * rng.normal(size=ref["weight"].shape) * live
    ((frag.vert_weight * t(g_w)).sum() + (frag.vert_hit_length * t(g_h)).sum()).backward()
    g_mu, g_sig = oracle_param_grads(ref, sig, g_w, g_h)
    if inverse:      # A = 2 / s: d/ds = -2 / s^2 d/dA (scalar)
        g_sig = -g_sig / (np.asarray(sig, np.float64) ** 2)
    grad_close(f"fragment_bwd K={K} B={B} {form} verts", n(gm.verts.grad), g_mu, 0.25 * TOL)
    grad_close(f"fragment_bwd K={K} B={B} {form} sigmas", n(gm.sigmas.grad), g_sig, 0.25 * TOL)
    assert np.abs(g_mu).max() > 0 and np.abs(g_sig).max() > 0


def test_silhouette_only_gradient_is_read_in_place(hip_lib, monkeypatch):
    """get_silhouette as the weights' only consumer: its backward hands ONE value per pixel, expanded over the slots
    with stride 0, and voge_fragment_bwd_iso reads that view in place (gw_stride_pix = 1, gw_stride_k = 0)."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import get_silhouette
    forbid_three_kernel_chain(monkeypatch)
    N, H, W, K = 1500, 48, 64, 25
    verts, sig, _ = random_scene(N, seed=11, lo=0.05, hi=0.1)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 30.0)
    renderer = renderer_for(H, W, K, 80.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    frag = renderer(gm, R=t(R), T=t(T))
    ref = oracle_frame(verts, sig, R, T, 80.0, (W / 2.0, H / 2.0), (H, W), K)
    same = same_lists(frag, ref, "silhouette only", max_flips=6)
    seen = {}
    real = ops._grad_weight_layout

    def spy(g, K_):
        out = real(g, K_)
        seen["strides"] = out[1:]
        return out
    monkeypatch.setattr(ops, "_grad_weight_layout", spy)
    tgt = np.random.default_rng(3).uniform(0, 1, same.shape)
    sil = get_silhouette(frag)
    (((sil - t(tgt)) ** 2) * t(same.astype(np.float32))).sum().backward()
    assert seen["strides"] == (1, 0), "the expanded per-pixel gradient must be read in place"
    wsum = ref["weight"].sum(-1)
    assert np.abs(n(sil) - np.minimum(wsum, 1))[same].max() < TOL
    g_pix = 2 * (np.minimum(wsum, 1) - tgt) * (wsum < 1) * same
    g_mu, g_sig = oracle_param_grads(ref, sig, np.broadcast_to(g_pix[..., None], ref["weight"].shape) *
                                     (np.arange(K)[None, None, None] < ref["valid_num"][..., None]))
    grad_close("silhouette-only verts", n(gm.verts.grad), g_mu, 0.25 * TOL)
    grad_close("silhouette-only sigmas", n(gm.sigmas.grad), g_sig, 0.25 * TOL)


# ----------------------------------------------------------------------------------------------- cfg5, the loop's ops
def test_config5_shapefit_iteration_interpolate_attr_and_silhouette(hip_lib, monkeypatch):
    """BASELINE config 5 at its real size with the loop's REAL op mix (demo/ShapeFitting.py:214-219,258-271,295):
    2562 Gaussians, 128x128, max_assign 25, five views rendered as ONE batch, image = interpolate_attr(frag, colours),
    silhouette = get_silhouette(frag), two MSE losses, one backward.  The one-pass backward must be the node that runs
    (the stand-alone chain is made unreachable) and the gradients of colours / verts / sigmas must match the oracle."""
    import importlib.util
    import os
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import get_silhouette, interpolate_attr
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("shape_fitting_demo", os.path.join(root, "demo", "ShapeFitting.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    v, _ = demo.ico_sphere(4)
    rng = np.random.default_rng(5)
    verts = (np.asarray(v, np.float32) * (1.0 + 0.05 * rng.normal(size=(2562, 1)))).astype(np.float32)
    sig = (np.full(2562, 1.0 / (0.05 ** 2 / (2 * np.log(1 / 0.6)))) * rng.uniform(0.8, 1.25, 2562)).astype(np.float32)
    cols = rng.uniform(0, 1, (2562, 3)).astype(np.float32)
    H = W = 128
    K, B = 25, 5
    R5, T5 = camera_np.look_at_view_transform([2.7] * 5, [0.0, 30.0, -20.0, 10.0, 45.0], [-180.0, -120.0, -40.0, 60.0, 160.0])
    forbid_three_kernel_chain(monkeypatch)
    renderer = renderer_for(H, W, K, 126.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    frag = renderer(gm, R=t(R5), T=t(T5))                     # five views, one call
    assert type(frag.vert_weight.grad_fn).__name__ in ("_FragmentsBackward", "_CompositeLeanBackward")
    img = interpolate_attr(frag, colors.repeat(B, 1))        # indices address rows b*N+n (RayTracing.py:24-30)
    sil = get_silhouette(frag)
    ref = oracle_frame(verts, sig, R5, T5, 126.0, (64.0, 64.0), (H, W), K)
    same = same_lists(frag, ref, "cfg5 5 views 128^2 K=25 (interpolate_attr + get_silhouette)", max_flips=12)
    colsB = np.tile(cols, (B, 1))
    rgb_ref = oracle.merge_fwd(colsB, ref["idx"], ref["weight"], ref["valid_num"])
    wsum = ref["weight"].sum(-1)
    assert np.abs(n(img) - rgb_ref)[same].max() < TOL and np.abs(n(sil) - np.minimum(wsum, 1))[same].max() < TOL
    assert (ref["valid_num"] > 0).mean() > 0.3
    tgt_rgb = rng.uniform(0, 1, rgb_ref.shape)
    tgt_sil = (rng.uniform(0, 1, wsum.shape) > 0.5).astype(np.float64)
    keep = t(same.astype(np.float32))
    loss = ((((img - t(tgt_rgb)) ** 2) * keep[..., None]).sum() / tgt_rgb[0].size
            + (((sil - t(tgt_sil)) ** 2) * keep).sum() / tgt_sil[0].size) / B
    loss.backward()
    g_rgb = 2 * (rgb_ref - tgt_rgb) / tgt_rgb[0].size / B * same[..., None]
    g_silh = 2 * (np.minimum(wsum, 1) - tgt_sil) / tgt_sil[0].size / B * same * (wsum < 1)
    g_attr, g_w = oracle.merge_bwd(colsB, ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    live = np.arange(K)[None, None, None] < ref["valid_num"][..., None]
    g_mu, g_sig = oracle_param_grads(ref, sig, g_w + g_silh[..., None] * live)
    want = (g_attr.reshape(B, 2562, 3).sum(0), g_mu, g_sig)
    for name, g, wv in zip(("colors", "verts", "sigmas"), (colors.grad, gm.verts.grad, gm.sigmas.grad), want):
        err = np.abs(n(g).astype(np.float64) - wv).max()
        print(f"[parity] cfg5 loop ops {name}: max err {err:.3e}, largest entry {np.abs(wv).max():.3e}")
        assert err <= 2 * TOL * np.abs(wv).max(), f"cfg5 {name}: {err:.3e} vs largest entry {np.abs(wv).max():.3e}"


# ------------------------------------------------------------------------------- EfficientCuboidViaOptimization's settings
@pytest.mark.parametrize("form", ["tril", "diag"])
def test_efficient_cuboid_settings_general_forms_vs_oracle(hip_lib, monkeypatch, form):
    """demo/EfficientCuboidViaOptimization.py:75-79,104-112: max_assign = the number of Gaussians (102), thr_activation = 0,
    max_point_per_bin = -1, sigmas = L L^T from a trainable lower-triangular L ([N,3,3] forms through GaussianRenderer),
    a SIX-channel attribute through interpolate_attr, L1 loss, gradients to the sigmas (and verts).  Also the (N,3)
    diagonal form (Aggregation.py:169-173).  Against the full oracle chain."""
    from voge_amd.Meshes import GaussianMeshesNaive
    from voge_amd.Renderer import interpolate_attr
    forbid_three_kernel_chain(monkeypatch)
    N, H, W, K = 102, 48, 64, 102
    rng = np.random.default_rng(17)
    # six faces of a cuboid, 17 Gaussians each, as the demo's efficient_cuboid() lays them out
    face = rng.integers(0, 6, N)
    verts = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    verts[np.arange(N), face % 3] = np.where(face < 3, 1.0, -1.0)
    if form == "tril":
        L0 = (np.eye(3)[None] * 2.0 + np.tril(rng.uniform(-0.4, 0.4, (N, 3, 3)))).astype(np.float32)
        L = t(L0, rg=True)
        sig_t = L @ L.transpose(1, 2)                                  # to_sym of the demo
    else:
        L0 = rng.uniform(2.0, 6.0, (N, 3)).astype(np.float32)
        L = t(L0, rg=True)
        sig_t = L
    sig = n(sig_t)
    verts_t = t(verts, rg=True)
    attr = np.eye(6, dtype=np.float32)[face]                          # idx_ of the demo: one-hot face labels, C = 6
    R, T = camera_np.look_at_view_transform(5.0, 25.0, 40.0)
    renderer = renderer_for(H, W, K, 60.0, thr=0.0)
    frag = renderer(GaussianMeshesNaive(verts_t, sig_t), R=t(R), T=t(T))
    ref = oracle_frame(verts, sig, R, T, 60.0, (W / 2.0, H / 2.0), (H, W), K, thr=0.0)
    same = same_lists(frag, ref, f"EfficientCuboid settings ({form})", max_flips=8)
    assert ref["valid_num"].max() > 32 and ref["valid_num"].mean() > 8      # thr_activation = 0: long lists, far from full
    img = interpolate_attr(frag, t(attr))
    rgb_ref = oracle.merge_fwd(attr, ref["idx"], ref["weight"], ref["valid_num"])
    assert np.abs(n(img) - rgb_ref)[same].max() < TOL
    tgt = rng.uniform(0, 1, rgb_ref.shape)
    (((img - t(tgt)).abs()) * t(same.astype(np.float32))[..., None]).sum().backward()      # L1Loss (sum-reduced)
    g_rgb = np.sign(rgb_ref - tgt) * same[..., None]
    _, g_w = oracle.merge_bwd(attr, ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    g_mu, g_sig = oracle_param_grads(ref, sig, g_w)
    grad_close(f"EfficientCuboid ({form}) verts", n(verts_t.grad), g_mu, 0.25 * TOL)
    if form == "tril":      # d/dL of L L^T: (G + G^T) L
        g_L = (g_sig + g_sig.transpose(0, 2, 1)) @ L0.astype(np.float64)
        grad_close("EfficientCuboid (tril) L", n(L.grad), g_L, 0.25 * TOL)
    else:
        grad_close("EfficientCuboid (diag) sigmas", n(L.grad), g_sig, 0.25 * TOL)


# ----------------------------------------------------------------------------------------------- host robustness
def test_frame_buffers_are_released_by_refcount(hip_lib):
    """nothing differentiable may hang on ctx as a plain attribute -- with the cyclic collector disabled the
    frame's tensors must die as soon as the last reference goes."""
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import get_silhouette, to_white_background
    verts, sig, cols = random_scene(800, seed=4, lo=0.05, hi=0.1)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 30.0)
    renderer = renderer_for(32, 32, 12, 50.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    gc.collect()
    gc.disable()
    try:
        for with_backward in (False, True):
            frag = renderer(gm, R=t(R), T=t(T))
            img = to_white_background(frag, colors)
            sil = get_silhouette(frag)
            refs = [weakref.ref(x) for x in (frag.vert_hit_length, frag.vert_weight, frag.vert_index, img, sil)]
            if with_backward:
                (img.sum() + sil.sum()).backward()
            del frag, img, sil
            assert all(r() is None for r in refs), [r() is None for r in refs]
    finally:
        gc.enable()


def test_fragment_views_keep_the_fast_paths(hip_lib, monkeypatch):
    """Fragments.copy() / squeeze() / unsqueeze() return views of the same memory (RenderBunny.py:45 renders
    to_white_background(frag.copy(), ...)): they must stay on the shade-through path and must not fall back to the
    synchronising idx.max() range check."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import interpolate_attr, to_white_background
    verts, sig, cols = random_scene(800, seed=5, lo=0.05, hi=0.1)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 30.0)
    renderer = renderer_for(32, 40, 12, 50.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)

    def no_sync(*a, **k):
        raise AssertionError("index range fell back to the device reduction")
    frag = renderer(gm, R=t(R), T=t(T))
    base = to_white_background(frag, colors)
    monkeypatch.setattr(torch.Tensor, "max", no_sync)
    for name, f in (("copy", frag.copy()), ("squeeze", frag.squeeze()), ("squeeze.unsqueeze", frag.squeeze().unsqueeze())):
        img = to_white_background(f, colors)
        assert type(img.grad_fn).__name__ in ("_ShadeThroughBackward", "_CompositeShadeBackward", "ViewBackward0"), (name, type(img.grad_fn).__name__)
        assert np.abs(n(img).reshape(tuple(base.shape)) - n(base)).max() <= 1e-6, name      # (one-pass vs shade-kernel sums)
        assert ops.hit_count_of(f.vert_index) is not None, name
        interpolate_attr(f, colors)
    monkeypatch.undo()
    # an in-place edit through torch invalidates the bookkeeping: the slow, checked path is taken again
    frag.vert_index[0, 0, 0, 0] = 0
    assert ops.hit_count_of(frag.vert_index) is None and ops.through_of(frag.vert_weight, frag.vert_index) is None


# ----------------------------------------------------------------------------------------------- closing self-comparisons
@pytest.mark.parametrize("form,K,lazy", [("full", 20, True), ("full", 33, True), ("diag", 12, True), ("full", 20, False),
                                         ("diag", 12, False), ("full", 20, "noad"), ("diag", 12, "noad")])
def test_shade_through_general_forms_vs_oracle(hip_lib, monkeypatch, form, K, lazy):
    """voge_fragment_shade_bwd (full 3x3 forms: to_colored_background on this renderer's fragments, ONE backward kernel)
    directly against the oracle chain -- round 2 compared it with this repo's own three kernels only.  lazy: the composite
    is deferred (voge_trace_lean_fwd keeps the packed (mu, A) records, voge_composite_shade_fwd_rec composites and shades in
    one pass, the backward re-derives act / dsd); otherwise the eager chain with act / dsd in memory (VOGE_LAZY_GENERAL=0)."""
    monkeypatch.setenv("VOGE_LAZY_GENERAL", "1" if lazy else "0")
    # ("noad": the deferred composite keeps no act / dsd either; the backward re-derives them from the packed (mu, A))
    monkeypatch.setenv("VOGE_GENERAL_KEEP_ACT_DSD", "0" if lazy == "noad" else "1")
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import to_colored_background
    N, H, W = 2000, 56, 72
    verts, sig, cols = random_scene(N, seed=500 + K, lo=0.05, hi=0.12, aniso=(form == "full"))
    if form == "diag":
        sig = (sig[:, None] * np.random.default_rng(K).uniform(0.6, 1.6, (N, 3))).astype(np.float32)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 30.0)
    bg = (0.9, 0.8, 1.0)
    renderer = renderer_for(H, W, K, 80.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    frag = renderer(gm, R=t(R), T=t(T))
    img = to_colored_background(frag, colors, background_color=bg)
    assert type(img.grad_fn).__name__ == ("_CompositeShadeBackward" if lazy else "_ShadeThroughBackward")
    ref = oracle_frame(verts, sig, R, T, 80.0, (W / 2.0, H / 2.0), (H, W), K)
    same = same_lists(frag, ref, f"shade-through {form} K={K}", max_flips=8)
    rgb = oracle.merge_fwd(cols, ref["idx"], ref["weight"], ref["valid_num"])
    img_ref, sil = oracle.blend_fwd(rgb, ref["weight"], bg)
    assert np.abs(n(img) - img_ref)[same].max() < TOL
    g_img = np.random.default_rng(1).normal(size=img_ref.shape) * same[..., None]
    (img * t(g_img)).sum().backward()
    x = rgb + (1 - sil)[..., None] * np.asarray(bg)
    g_rgb = g_img * (x < 1)
    g_sumw = -(g_rgb * np.asarray(bg)).sum(-1) * (ref["weight"].sum(-1) < 1)
    g_attr, g_w = oracle.merge_bwd(cols, ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    live = np.arange(K)[None, None, None] < ref["valid_num"][..., None]
    g_mu, g_sig = oracle_param_grads(ref, sig, g_w + g_sumw[..., None] * live)
    grad_close(f"shade-through {form} K={K} colors", n(colors.grad), g_attr, 0.25 * TOL)
    grad_close(f"shade-through {form} K={K} verts", n(gm.verts.grad), g_mu, 0.25 * TOL)
    grad_close(f"shade-through {form} K={K} sigmas", n(gm.sigmas.grad), g_sig, 0.25 * TOL)


def test_ray_kernel_against_the_reference_get_ray_camera_space(hip_lib):
    """voge_rays_fwd against the fixture the reference's own get_ray_camera_space produced (VoGE/Aggregation.py:11-27,
    tests/golden/make_golden.py: gen_ray_camera_space): identical view-space directions once the principal point is
    moved by the half pixel that separates pixel corners (the reference's helper) from pixel centres (PyTorch3D's
    sampler, which the renderer uses)."""
    import os
    from util import GOLDEN
    from voge_amd.cameras import PerspectiveCameras, pixel_rays
    g = np.load(os.path.join(GOLDEN, "ray_camera_space.npz"))
    for name in "abc":
        H, W = (int(v) for v in g[name + "_size"])
        py, px = g[name + "_principle_row_col"]
        fy, fx = g[name + "_focal_row_col"]
        cams = PerspectiveCameras(focal_length=((float(fx), float(fy)),), principal_point=((float(px) + 0.5, float(py) + 0.5),),
                                  image_size=((H, W),), device=DEV)
        rays, origin = pixel_rays(cams, (H, W))
        assert origin.abs().max().item() == 0
        assert np.abs(n(rays)[0] - g[name + "_dirs"]).max() < 5e-7, name


# ----------------------------------------------------------------------------------------------- deferred composite
@pytest.mark.parametrize("K,B,C,inverse", [(40, 1, 3, False), (20, 2, 4, False), (8, 1, 3, True), (128, 1, 3, False)])
def test_deferred_composite_equals_the_eager_chain(hip_lib, monkeypatch, K, B, C, inverse):
    """Scalar-sigma fragments come back with their composite deferred (ops.LAZY_COMPOSITE): to_colored_background then
    produces weights and image in one pass (voge_composite_shade_fwd_iso).  Against the eager chain (trace + composite in
    the renderer, shade kernel afterwards): identical index lists / hit lengths / valid_num / weights, images within 1e-6,
    the same gradients -- with a silhouette loss and a hit-length loss on the same fragments, whose gradients reach the
    weights of the deferred node from outside."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import get_silhouette, to_colored_background
    N, H, W = 2500, 60, 76
    verts, sig, cols = random_scene(N, seed=700 + K, lo=0.05, hi=0.12)
    if inverse:
        sig = (1.0 / sig).astype(np.float32)
    cols = np.concatenate([cols, cols[:, :1]], axis=1)[:, :C]
    R, T = camera_np.look_at_view_transform([3.0, 3.3][:B], [10.0, -20.0][:B], [30.0, 200.0][:B])
    renderer = renderer_for(H, W, K, 85.0, occ=1.2, inverse=inverse)
    gen = torch.Generator(DEV).manual_seed(5)
    g_img = torch.randn(B, H, W, C, device=DEV, generator=gen)
    g_sil = torch.randn(B, H, W, device=DEV, generator=gen)
    bg = tuple([0.9, 0.8, 1.0, 0.7][:C])
    out = {}
    for lazy in (True, False):
        monkeypatch.setattr(ops, "LAZY_COMPOSITE", lazy)
        gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
        colors = t(np.tile(cols, (B, 1)), rg=True)
        frag = renderer(gm, R=t(R), T=t(T))
        assert (frag._lazy is not None) == lazy
        img = to_colored_background(frag, colors, background_color=bg)
        assert type(img.grad_fn).__name__ == ("_CompositeShadeBackward" if lazy else "_ShadeThroughBackward")
        assert frag._lazy is None                                  # the fragments are complete now
        hl = torch.where(frag.vert_index >= 0, frag.vert_hit_length, torch.zeros_like(frag.vert_hit_length))
        ((img * g_img).sum() + (get_silhouette(frag) * g_sil).sum() + 0.01 * hl.sum()).backward()
        out[lazy] = [n(x) for x in (frag.vert_index, frag.vert_hit_length, frag.valid_num, frag.vert_weight, img, gm.verts.grad,
                                    gm.sigmas.grad, colors.grad)]
    for a, b in zip(out[True][:4], out[False][:4]):
        assert np.array_equal(a, b)
    assert np.abs(out[True][4] - out[False][4]).max() <= 1e-6
    for name, a, b in zip(("verts", "sigmas", "colors"), out[True][5:], out[False][5:]):
        grad_close("deferred vs eager composite, " + name, a, b, 0.25 * TOL)


def test_deferred_composite_materialises_on_first_read(hip_lib):
    """Reading vert_weight / valid_num (interpolate_attr, get_silhouette, a plain attribute access) runs the composite
    then; a later to_white_background takes the shade-through path on the finished fragments.  Everything is what the
    eager renderer gives."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import interpolate_attr, to_white_background
    verts, sig, cols = random_scene(1500, seed=8, lo=0.05, hi=0.1)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 30.0)
    renderer = renderer_for(48, 64, 16, 70.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    frag = renderer(gm, R=t(R), T=t(T))
    assert frag._lazy is not None and len(frag) == 1 and frag.vert_index.shape == (1, 48, 64, 16)
    vn = frag.valid_num                                             # the hit count: no composite needed
    assert frag._lazy is not None and vn.dtype == torch.int64
    cp = frag.copy()
    assert cp._lazy is frag._lazy
    rgb = interpolate_attr(frag, colors)                            # composite + merge (+ weight sum) in one pass, now
    assert frag._lazy is None and type(frag.vert_weight.grad_fn).__name__ == "_CompositeMergeBackward"
    assert torch.equal(frag.valid_num, vn)
    f3 = renderer(gm, R=t(R), T=t(T))
    w3 = f3.vert_weight                                             # a plain read: the composite kernel alone
    assert f3._lazy is None and type(w3.grad_fn).__name__ == "_CompositeLeanBackward" and torch.equal(w3, frag.vert_weight)
    assert (interpolate_attr(f3, colors) - rgb).abs().max().item() <= 1e-6
    from voge_amd.Renderer import get_silhouette
    assert (get_silhouette(frag) - get_silhouette(f3)).abs().max().item() <= 1e-6      # cached weight sum vs the kernel
    img = to_white_background(frag, colors)
    assert type(img.grad_fn).__name__ == "_ShadeThroughBackward"
    img2 = to_white_background(cp, colors)                          # the copy is still deferred: one-pass form
    assert type(img2.grad_fn).__name__ == "_CompositeShadeBackward" and (img2 - img).abs().max().item() <= 1e-6
    assert torch.equal(cp.vert_weight, frag.vert_weight)
    with torch.no_grad():
        f2 = renderer(gm, R=t(R), T=t(T))
        assert f2._lazy is not None
        assert (to_white_background(f2, colors) - img).abs().max().item() <= 1e-6
    (rgb.sum() + img.sum()).backward()
    assert torch.isfinite(gm.verts.grad).all() and gm.verts.grad.abs().max() > 0


@pytest.mark.parametrize("K,B,C", [(25, 2, 3), (40, 1, 4), (7, 1, 3)])
def test_training_pattern_one_pass_vs_oracle(hip_lib, monkeypatch, K, B, C):
    """interpolate_attr + get_silhouette on deferred-composite fragments: ONE forward kernel behind the sweep
    (voge_composite_shade_fwd_iso without a background: weights, merged attributes, weight sum) and ONE backward kernel
    (voge_fragment_merge_bwd_iso) -- any K, ShapeFitting's 25 included.  Forward and gradients against the oracle chain;
    the stand-alone merge / silhouette / composite / trace kernels are made unreachable."""
    from voge_amd import _lib
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import get_silhouette, interpolate_attr
    forbid_three_kernel_chain(monkeypatch)
    lib = _lib.load()

    def boom(*a):
        raise AssertionError("a stand-alone stage kernel was launched")
    for name in ("voge_merge_fwd", "voge_merge_bwd", "voge_silhouette_fwd", "voge_silhouette_bwd", "voge_composite_fwd_iso", "voge_fragment_bwd_iso"):
        monkeypatch.setattr(lib, name, boom, raising=True)
    N, H, W = 2200, 52, 68
    verts, sig, cols = random_scene(N, seed=900 + K, lo=0.05, hi=0.12)
    cols = np.concatenate([cols, cols[:, :1]], axis=1)[:, :C]
    R, T = camera_np.look_at_view_transform([3.0, 3.3][:B], [10.0, -20.0][:B], [30.0, 200.0][:B])
    renderer = renderer_for(H, W, K, 85.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    frag = renderer(gm, R=t(R), T=t(T))
    rgb = interpolate_attr(frag, colors.repeat(B, 1))
    sil = get_silhouette(frag)
    assert type(rgb.grad_fn).__name__ == "_CompositeMergeBackward"
    ref = oracle_frame(verts, sig, R, T, 85.0, (W / 2.0, H / 2.0), (H, W), K)
    same = same_lists(frag, ref, f"training pattern, one pass K={K} B={B} C={C}", max_flips=10)
    colsB = np.tile(cols, (B, 1))
    rgb_ref = oracle.merge_fwd(colsB, ref["idx"], ref["weight"], ref["valid_num"])
    wsum = ref["weight"].sum(-1)
    assert np.abs(n(rgb) - rgb_ref)[same].max() < TOL and np.abs(n(sil) - np.minimum(wsum, 1))[same].max() < TOL
    rng = np.random.default_rng(K)
    g_rgb = rng.normal(size=rgb_ref.shape) * same[..., None]
    g_silh = rng.normal(size=wsum.shape) * same
    ((rgb * t(g_rgb)).sum() + (sil * t(g_silh)).sum()).backward()
    g_attr, g_w = oracle.merge_bwd(colsB, ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    live = np.arange(K)[None, None, None] < ref["valid_num"][..., None]
    g_mu, g_sig = oracle_param_grads(ref, sig, g_w + (g_silh * (wsum < 1))[..., None] * live)
    grad_close(f"one-pass training pattern K={K} colors", n(colors.grad), g_attr.reshape(B, N, C).sum(0), 0.25 * TOL)
    grad_close(f"one-pass training pattern K={K} verts", n(gm.verts.grad), g_mu, 0.25 * TOL)
    grad_close(f"one-pass training pattern K={K} sigmas", n(gm.sigmas.grad), g_sig, 0.25 * TOL)


def test_replayed_training_iteration_equals_eager(hip_lib):
    """demo/ShapeFitting.py BatchedIteration captured into a HIP graph and replayed back to back (no host
    synchronisation) must follow the eagerly launched loop step for step.  Regression test of a round-3 finding: a
    hipMemsetAsync inside the captured backward did not take effect on replay, the accumulators kept their previous contents
    and ONE replayed SGD step sent every vertex to 1e20 -- the library zeroes with a kernel of its own since."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("shape_fitting_demo", os.path.join(root, "demo", "ShapeFitting.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    runs = {}
    for graph in (False, True):
        h = demo.fit(iters=40, level=3, size=64, max_assign=13, rgb_on=10, quiet=True, graph=graph)
        runs[graph] = h
    for key in ("silhouette", "rgb"):
        a, b = np.asarray(runs[True][key]), np.asarray(runs[False][key])
        assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max()), (key, a[:6], b[:6])
    assert np.abs(runs[True]["final_verts"] - runs[False]["final_verts"]).max() < 1e-3
    assert np.asarray(runs[True]["silhouette"])[-1] < 0.8 * np.asarray(runs[True]["silhouette"])[0]


@pytest.mark.parametrize("kind,shared", [("diag", True), ("full", True), ("diag", False), ("full", False)])
def test_general_preamble_equals_the_torch_chain(hip_lib, kind, shared):
    """voge_general_preamble_fwd / _bwd against the reference's own operations (Renderer.py:130-137: centred = verts -
    origin[b], isigma = 2 * expend_sigma(sigmas)) and their autograd: identical values (the same fp32 operations),
    gradients to rounding of the sum over views."""
    from voge_amd import ops
    from voge_amd.Aggregation import expend_sigma
    B, N = 3, 257
    g = torch.Generator().manual_seed(5)
    vshape = (N, 3) if shared else (B, N, 3)
    sshape = ((N, 3) if kind == "diag" else (N, 3, 3)) if shared else ((B, N, 3) if kind == "diag" else (B, N, 3, 3))
    verts = torch.randn(vshape, generator=g).to(DEV).requires_grad_(True)
    sigmas = (torch.rand(sshape, generator=g) + 0.5).to(DEV).requires_grad_(True)
    origin = torch.randn((B, 3), generator=g).to(DEV)
    mus, isg = ops.general_preamble(verts, sigmas, origin)
    v2, s2 = verts.detach().clone().requires_grad_(True), sigmas.detach().clone().requires_grad_(True)
    vb = v2[None].expand(B, -1, -1) if shared else v2
    centred = (vb - origin[:, None]).reshape(-1, 3)
    if shared:
        sig3 = expend_sigma(s2).unsqueeze(0).expand(B, -1, -1, -1)
    else:
        sig3 = torch.stack([expend_sigma(s2[b]) for b in range(B)])
    isg_ref = (2 * sig3).reshape(-1, 3, 3)
    assert torch.equal(mus, centred) and torch.equal(isg, isg_ref)
    gm, ga = torch.randn(mus.shape, generator=g).to(DEV), torch.randn(isg.shape, generator=g).to(DEV)
    ((mus * gm).sum() + (isg * ga).sum()).backward()
    ((centred * gm).sum() + (isg_ref * ga).sum()).backward()
    assert (verts.grad - v2.grad).abs().max().item() <= 1e-5 * max(1.0, v2.grad.abs().max().item())
    assert (sigmas.grad - s2.grad).abs().max().item() <= 1e-5 * max(1.0, s2.grad.abs().max().item())


@pytest.mark.parametrize("form", ["scalar", "full"])
def test_graph_replay_of_a_small_dense_object(hip_lib, form):
    """A frame whose quads take binB's pooled long path (and whose segments use binA's extensions), captured into a HIP
    graph: the pool counter is reset by binA on every replay and the extension arenas by an LDS counter, so back-to-back
    replays must reproduce the eager frame's image and gradients.  Scalar and full 3x3 forms (deferred composite)."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import to_white_background
    N, H, W, K = 30000, 96, 96, 16
    rng = np.random.default_rng(11)
    verts = rng.uniform(-0.3, 0.3, (N, 3)).astype(np.float32)
    r = rng.uniform(0.006, 0.012, N)
    s = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
    sig = s if form == "scalar" else (s[:, None, None] * np.eye(3, dtype=np.float32)[None] * rng.uniform(0.7, 1.4, (N, 3, 1))).astype(np.float32)
    cols = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 20.0)
    renderer = renderer_for(H, W, K, 110.0)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    Rt, Tt = t(R), t(T)
    params = [gm.verts, gm.sigmas, colors]

    def step():
        for p in params:
            p.grad = None
        img = to_white_background(renderer(gm, R=Rt, T=Tt), colors)
        img.sum().backward()
        return img
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            img_e = step()
        used, cap = ops.trace_pool_usage(side.device, 1, N, H, W)      # (the workspace of THIS stream)
    assert 0 < used <= cap, (used, cap)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want_img = img_e.detach().clone()
    want = [p.grad.detach().clone() for p in params]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        img_g = step()
    for _ in range(4):
        graph.replay()
    torch.cuda.synchronize()
    assert (img_g - want_img).abs().max().item() < 1e-6
    for p, w in zip(params, want):
        assert (p.grad - w).abs().max().item() <= 2e-4 * max(1.0, w.abs().max().item())      # (atomics: order of the sums)
    assert float(want_img.min()) < 0.9      # the object is there


# ------------------------------------------------------------------ the deferred composite and the autograd mode
@pytest.mark.parametrize("sig_kind", ["scalar", "full3x3"])
def test_deferred_composite_keeps_the_render_calls_autograd_mode(sig_kind):
    """The reference composites inside the renderer call (Renderer.py:139-143), so WHEN somebody first reads the weights
    cannot matter.  Here the composite is deferred: it must run under the autograd mode of the render call, not of the
    first reader.  (a) render with grad, read vert_weight / valid_num under no_grad (a feature-bank update, logging), then
    differentiate a loss through interpolate_attr: the Gaussians still get their gradient, and it equals the one of the
    same frame whose weights were never touched under no_grad.  (b) fragments rendered under no_grad stay graph-free when
    they are composited outside it.  (c) assigning vert_weight before anything was read leaves valid_num in place."""
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import interpolate_attr, to_white_background
    verts, sig, cols = random_scene(500, seed=11, lo=0.07, hi=0.16)
    if sig_kind == "full3x3":
        sig = camera_np.expand_sigma(sig).astype(np.float32)
    H, W, K = 48, 64, 12
    R, T = camera_np.look_at_view_transform(3.2, 15.0, 35.0)
    renderer = renderer_for(H, W, K, 60.0)

    def frame(read_first):
        gm = GaussianMeshes(t(verts, rg=True), t(sig, rg=True))
        colors = t(cols, rg=True)
        frag = renderer(gm, R=t(R), T=t(T))
        if read_first:
            with torch.no_grad():
                w = frag.vert_weight
                assert frag.valid_num is not None and w.shape[-1] == K
                _ = frag.shape, frag[0]
        img = interpolate_attr(frag, colors)
        (img * t(np.linspace(0.5, 1.5, 3))).sum().backward()
        return n(gm.verts.grad), n(gm.sigmas.grad), n(colors.grad)

    plain, touched = frame(False), frame(True)
    for name, a, b in zip(("verts", "sigmas", "colors"), plain, touched):
        assert np.abs(a).max() > 0, name
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max()), name      # (atomics order: not bit-equal)

    gm = GaussianMeshes(t(verts, rg=True), t(sig, rg=True))
    with torch.no_grad():
        frag = renderer(gm, R=t(R), T=t(T))
    w = frag.vert_weight                      # composited outside no_grad: still no graph, nothing kept for a backward
    assert not w.requires_grad and w.grad_fn is None
    img = to_white_background(frag, t(cols))
    assert not img.requires_grad

    frag = renderer(gm, R=t(R), T=t(T))
    frag.vert_weight = torch.zeros_like(frag.vert_hit_length)
    assert frag.valid_num is not None and frag.valid_num.shape == frag.vert_index.shape[:-1]


def test_every_public_fragments_transformation_is_classified(hip_lib):
    """the fast paths hang on Python attributes of the fragment tensors (ops.carry_tags), so every public way
    to get Fragments out of Fragments must either KEEP them or be LISTED as dropping them -- a new method that is neither
    fails here.  KEEP: the result still takes the one-pass backward / needs no synchronising range check, with the
    composite deferred or done.  DROP (documented): a true slice of a multi-view batch is other
    memory -- it takes the checked, three-kernel paths with the same values."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import Fragments
    KEEP = {"copy": lambda f: f.copy(), "squeeze": lambda f: f.squeeze(), "unsqueeze": lambda f: f.squeeze().unsqueeze(),
            "__getitem__": lambda f: f[0]}
    DROP = {"__getitem__ of a multi-view batch": None}
    NOT_A_TRANSFORMATION = {"to_dict", "shape", "vert_weight", "valid_num", "vert_hit_length", "__len__", "__init__"}
    public = {k for k, v in vars(Fragments).items() if not k.startswith("_") or k in ("__getitem__", "__len__", "__init__")}
    public -= {"_fields"}
    assert public == set(KEEP) | NOT_A_TRANSFORMATION, public ^ (set(KEEP) | NOT_A_TRANSFORMATION)
    verts, sig, cols = random_scene(600, seed=9, lo=0.06, hi=0.12)
    R1, T1 = camera_np.look_at_view_transform(3.0, 10.0, 30.0)
    renderer = renderer_for(32, 40, 10, 50.0)
    for deferred in (True, False):
        gm = GaussianMeshes(t(verts, rg=True), t(sig, rg=True))
        for name, fn in KEEP.items():
            frag = renderer(gm, R=t(R1), T=t(T1))
            if not deferred:
                _ = frag.vert_weight                      # composite now: the tags live on the weight tensor
            g = fn(frag)
            assert isinstance(g, Fragments), name
            assert ops.hit_count_of(g.vert_index) is not None, (name, deferred)
            assert getattr(g.vert_index, "voge_index_bound", None) is not None, (name, deferred)
            w = g.vert_weight
            assert ops.through_of(w, g.vert_index) is not None, (name, deferred)      # -> the one-pass backward
    # the documented drop: one view of a two-view batch
    R2, T2 = camera_np.look_at_view_transform([3.0, 3.2], [10.0, -5.0], [30.0, 100.0])
    gm = GaussianMeshes(t(verts, rg=True), t(sig, rg=True))
    frag = renderer(gm, R=t(R2), T=t(T2))
    one = frag[1]
    assert ops.hit_count_of(one.vert_index) is None and ops.through_of(one.vert_weight, one.vert_index) is None
    assert torch.equal(one.vert_weight, frag.vert_weight[1]) and list(DROP)


def test_duplicate_ids_in_an_edited_list_take_the_elected_kernels(hip_lib, monkeypatch):
    """the one-pass backward adds a pixel's slots to its table without arbitration -- safe for the lists the trace
    wrote (a Gaussian sits in a pixel's list once), a race for a list somebody edited so that it holds a Gaussian TWICE.
    _Fragments.backward sees the edit (the index tensor's version counter) and runs the stand-alone kernels, whose accumulation
    elects one writer per key: gradients against the oracle chain on the edited list, duplicates summed."""
    from voge_amd import ops
    monkeypatch.setenv("VOGE_FRAGMENTS_KEEP_ACT_DSD", "1")      # (the reference's layout: act / dsd as the forward wrote them)
    verts, sig, _ = random_scene(700, seed=77, lo=0.06, hi=0.14)
    H, W, K = 32, 40, 10
    R, T = camera_np.look_at_view_transform(3.1, 5.0, 25.0)
    rays, origin = camera_np.pixel_rays(R, T, 42.0, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32).reshape(-1, 3)
    a = (2 * sig).astype(np.float32)
    tm, ta = t(mus, rg=True), t(a, rg=True)
    thr_act = oracle.thr_act_of(0.01)
    weight, idx, valid, ln = ops.fragments(1, tm, ta, None, t(rays), None, thr_act, K)
    th = weight.voge_through
    act0, dsd0, len0, idx0 = (n(th[k]).copy() for k in ("act", "dsd", "len", "idx"))
    full = n(valid)[0] >= 3
    assert full.sum() > 50
    with torch.no_grad():      # slot 1 := slot 0's Gaussian wherever a pixel holds three or more: the same id twice per pixel
        idx[0, :, :, 1] = torch.where(torch.from_numpy(full).to(DEV), idx[0, :, :, 0], idx[0, :, :, 1])
    g_w = np.random.default_rng(5).normal(size=act0.shape)
    (weight * t(g_w)).sum().backward()
    g_act, g_len, g_dsd = oracle.composite_bwd(act0, len0, dsd0, g_w * (idx0 >= 0), 1.0)
    isg = (a[:, None, None] * np.eye(3, dtype=np.float32)[None]).astype(np.float32)
    _, g_mu, g_A = oracle.trace_bwd(mus, isg, rays, n(idx), g_len, g_act, g_dsd)
    grad_close("duplicate ids: means", n(tm.grad), g_mu, TOL)
    grad_close("duplicate ids: a", n(ta.grad), np.einsum("nii->n", g_A), TOL)
