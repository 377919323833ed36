"""Round 6: the frame path of ABI 7 -- the renderer's trace with the CAMERA as input (voge_frame_trace_fwd_iso: no
ray-generation launch; rays, cones, camera centre and view axis made inside binA / binB / the sweep), the composite that
zeroes the backward's gradient arrays on its way (voge_frame_shade_fwd_iso) and the one-launch backward that adds straight
into them (voge_frame_shade_bwd_iso).

What is asserted:
  * the ray bundle the sweep leaves behind and the camera centre are the bits of voge_rays_fwd (Renderer.py:124-130's bundle),
    for whole frames, row bands, interleaved stripes, ragged sizes and batches;
  * fragments and image are the bits of round 5's chain (rays kernel + rays-taking entries) -- the analytic corner-ray cones
    cull conservatively, so the exact sweep sees a different candidate ORDER and must return the same lists;
  * the direct-flush backward equals the acc + fill + finish form up to the order of its float atomics, on every sigma rule,
    for shared sets over a batch, and a second backward over the same graph (which takes the scratch form) equals the first;
  * against the fp64 oracle on a frame of its own.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import camera_np
from util import TOL, log_line

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a, dtype=torch.float32, rg=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=rg)


def n(x):
    return x.detach().cpu().numpy()


def _scene(N, seed, lo=0.03, hi=0.08):
    from voge_amd import scenes
    return scenes.random_gaussians(N, seed=seed, r_lo=lo, r_hi=hi)


def _views(B, seed):
    rng = np.random.default_rng(seed)
    return camera_np.look_at_view_transform(list(rng.uniform(3.0, 4.0, B)), list(rng.uniform(-30, 30, B)), list(rng.uniform(-180, 180, B)))


def _render(verts, sig, cols, R, T, size, K, frame, focal=None, pp=None, rows=None, bins=-1, inverse_sigma=False, grad=True, thr=-1):
    """One forward (+ the tensors to differentiate) through the public API with the frame path on or off."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_colored_background
    from voge_amd.cameras import PerspectiveCameras
    H, W = size
    focal = float(1.2 * max(H, W)) if focal is None else focal
    pp = (W / 2.0 + 0.25, H / 2.0 - 0.5) if pp is None else pp
    cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=(size,), device=DEV)
    st = GaussianRenderSettings(image_size=size, max_assign=K, max_point_per_bin=bins, inverse_sigma=inverse_sigma)
    renderer = GaussianRenderer(cams, st).to(DEV)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=grad)
    B = np.asarray(R).reshape(-1, 3, 3).shape[0]
    old = ops.FRAME_PATH
    ops.FRAME_PATH = frame
    try:
        frag = renderer(gm, R=t(R), T=t(T), **({} if rows is None else dict(rows=rows)))
        lz = frag._lazy
        assert (lz is not None and lz.frame) == frame
        img = to_colored_background(frag, colors.repeat(B, 1) if B > 1 else colors, (0.9, 1.0, 0.8), thr=thr)
    finally:
        ops.FRAME_PATH = old
    return frag, img, gm, colors, lz


@pytest.mark.parametrize("B,size,rows,what", [
    (1, (64, 64), None, "a whole frame"),
    (2, (70, 53), None, "ragged sizes (partial tiles, quads and super-tiles), a batch of two"),
    (1, (256, 256), (100, 164), "a band of 64 rows"),
    (1, (96, 200), (37, 90), "a band that starts and ends inside a tile"),
    (3, (33, 31), None, "one super-tile and a bit, three views"),
])
def test_sweep_leaves_the_ray_kernels_bundle(hip_lib, B, size, rows, what):
    """rays [B,h,W,3] written by the sweep (every tile, the empty ones too) and origin [B,3] written by the first kernel
    == voge_rays_fwd's, bit for bit (csrc/voge_common.h: cam_ray / cam_origin are what rays_fwd_kernel is built from)."""
    from voge_amd import ops
    from voge_amd.cameras import PerspectiveCameras, camera_tensors, pixel_rays
    verts, sig, cols = _scene(700, seed=B * 7 + size[0])
    R, T = _views(B, seed=size[1])
    H, W = size
    cams = PerspectiveCameras(focal_length=1.3 * max(H, W), principal_point=((W / 2.0 - 0.75, H / 2.0 + 0.125),), image_size=(size,),
                              R=t(R), T=t(T), device=DEV)
    want_rays, want_origin = pixel_rays(cams, size, rows=rows)
    cam = camera_tensors(cams, size, rows)
    assert cam is not None
    origin = torch.empty((B, 3), device=DEV)
    idx, ln, lz = ops.frame_trace(t(verts), t(sig), *cam[:4], cam[4], cam[5], False, oracle.thr_act_of(0.01), 12, 1, 1.0, origin_out=origin)
    assert lz.rays.shape == want_rays.shape, what
    assert torch.equal(lz.rays, want_rays), what
    assert torch.equal(origin, want_origin), what


def test_sweep_leaves_the_bundle_of_interleaved_stripes(hip_lib):
    """A rank's interleaved stripes, stacked (voge_rays_striped_fwd's row rule): same bits; 16-row stripes put two stripes into
    every super-tile, whose analytic cone then bounds the rows between them as well (a superset: still conservative)."""
    from voge_amd import ops
    from voge_amd.cameras import PerspectiveCameras, camera_tensors, pixel_rays
    from voge_amd.distributed import Stripes
    verts, sig, cols = _scene(3000, seed=5)
    R, T = camera_np.look_at_view_transform(3.2, 20.0, -35.0)
    size = (256, 192)
    cams = PerspectiveCameras(focal_length=260.0, principal_point=((96.0, 128.0),), image_size=(size,), R=t(R), T=t(T), device=DEV)
    for stripe_h, world, rank in ((32, 4, 1), (16, 2, 1), (32, 8, 7)):
        st = Stripes(size[0], rank, world, stripe_h)
        want_rays, want_origin = pixel_rays(cams, size, rows=st)
        cam = camera_tensors(cams, size, st)
        origin = torch.empty((1, 3), device=DEV)
        idx, ln, lz = ops.frame_trace(t(verts), t(sig), *cam[:4], cam[4], cam[5], False, oracle.thr_act_of(0.01), 16, 1, 1.0, origin_out=origin)
        assert torch.equal(lz.rays, want_rays) and torch.equal(origin, want_origin)
        # ... and the stacked band's fragments are round 5's (rays-taking entries on the same stacked bundle)
        idx0, ln0, lz0 = ops.trace_lean(2, t(verts), t(sig), want_origin, want_rays, None, oracle.thr_act_of(0.01), 16, 1, 1.0)
        assert torch.equal(idx, idx0) and torch.equal(ln, ln0) and torch.equal(lz.cnt, lz0.cnt)


@pytest.mark.parametrize("N,B,size,K,bins,what", [
    (866, 1, (128, 128), 20, -1, "a few hundred Gaussians: the small-set path (no binA)"),
    (6000, 2, (96, 120), 16, -1, "binA derives the records; a batch of two"),
    (6000, 1, (96, 120), 16, None, "default bins: the view-axis rule from column 2 of R"),
    (140000, 1, (64, 64), 8, -1, "more than 131 072 Gaussians: the records in a pass of their own"),
    (50000, 1, (512, 512), 40, -1, "the headline config's size"),
])
def test_frame_path_equals_the_ray_bundle_chain(hip_lib, N, B, size, K, bins, what):
    """Fragments, image and every stage's bookkeeping of the frame path == round 5's chain, bit for bit (the forward has no
    atomics); the gradients agree to the order of the backward's float atomics."""
    verts, sig, cols = _scene(N, seed=N % 97, lo=0.02 if N >= 50000 else 0.04, hi=0.04 if N >= 50000 else 0.09)
    R, T = _views(B, seed=N % 13)
    a = _render(verts, sig, cols, R, T, size, K, frame=True, bins=bins)
    b = _render(verts, sig, cols, R, T, size, K, frame=False, bins=bins)
    assert torch.equal(a[4].rays, b[4].rays), what
    for name in ("vert_index", "vert_hit_length", "valid_num", "vert_weight"):
        assert torch.equal(getattr(a[0], name), getattr(b[0], name)), (what, name)
    assert torch.equal(a[4].cnt, b[4].cnt) and torch.equal(a[4].records, b[4].records), what
    assert torch.equal(a[1], b[1]), what
    g_img = t(np.random.default_rng(N).normal(size=tuple(a[1].shape)))
    for r in (a, b):
        (r[1] * g_img).sum().backward()
    worst = 0.0
    for name, ga, gb in (("colors", a[3].grad, b[3].grad), ("verts", a[2].verts.grad, b[2].verts.grad), ("sigmas", a[2].sigmas.grad, b[2].sigmas.grad)):
        scale = max(1.0, float(gb.abs().max()))
        err = float((ga - gb).abs().max()) / scale
        worst = max(worst, err)
        assert err < 2e-5, (what, name, err)
    log_line(f"[frame] {what}: frame path == ray-bundle chain bit for bit forward; gradients within {worst:.1e} of scale (atomics order)")


@pytest.mark.parametrize("inverse_sigma,B", [(False, 1), (True, 1), (False, 3), (True, 2)])
def test_direct_flush_backward_sigma_rules_and_shared_sets(hip_lib, inverse_sigma, B):
    """voge_frame_shade_bwd_iso folds the sigma rule's chain factor (2, or -2 / sigma^2) and the sum over the views of a shared
    Gaussian set into its flush; against the acc + fill + finish form (VOGE_FRAME_DIRECT_BWD=0's path) on the same fragments."""
    from voge_amd import ops
    verts, sig, cols = _scene(1500, seed=11 + B)
    if inverse_sigma:
        sig = (1.0 / sig).astype(np.float32)
    R, T = _views(B, seed=3 * B)
    out = []
    for direct in (True, False):
        old = ops.FRAME_DIRECT_BWD
        ops.FRAME_DIRECT_BWD = direct
        try:
            r = _render(verts, sig, cols, R, T, (80, 72), 14, frame=True, inverse_sigma=inverse_sigma)
        finally:
            ops.FRAME_DIRECT_BWD = old
        g_img = t(np.random.default_rng(9).normal(size=tuple(r[1].shape)))
        (r[1] * g_img).sum().backward()
        out.append((r[3].grad, r[2].verts.grad, r[2].sigmas.grad))
    for name, ga, gb in zip(("colors", "verts", "sigmas"), out[0], out[1]):
        scale = max(1.0, float(gb.abs().max()))
        assert float((ga - gb).abs().max()) / scale < 2e-5, (name, inverse_sigma, B)
        assert float(gb.abs().max()) > 0


def test_second_backward_over_the_same_graph(hip_lib):
    """The zeroed gradient buffer serves ONE backward; a second one over the retained graph takes the scratch form and must
    return the same gradients (and must not have been corrupted by the first's accumulation)."""
    verts, sig, cols = _scene(2000, seed=21)
    R, T = _views(1, seed=4)
    frag, img, gm, colors, lz = _render(verts, sig, cols, R, T, (64, 96), 12, frame=True)
    loss = (img * img).sum()
    g1 = torch.autograd.grad(loss, (colors, gm.verts, gm.sigmas), retain_graph=True)
    g2 = torch.autograd.grad(loss, (colors, gm.verts, gm.sigmas))
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_frame_path_against_the_oracle(hip_lib):
    """The frame path on a frame of its own (ragged size, off-centre principal point, threshold silhouette off) against the
    fp64 oracle chain: fragments, image and the three gradients."""
    import test_gpu_configs as C
    verts, sig, cols = _scene(1800, seed=33)
    R, T = camera_np.look_at_view_transform(3.4, -12.0, 55.0)
    size = (90, 75)
    sc = dict(verts=verts, sigmas=sig, colors=cols, focal=140.0, principal=(40.0, 41.5), image_size=size, K=18)
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    cams = PerspectiveCameras(focal_length=140.0, principal_point=((40.0, 41.5),), image_size=(size,), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=size, max_assign=18, max_point_per_bin=-1)).to(DEV)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    frag = renderer(gm, R=t(R), T=t(T))
    assert frag._lazy is not None and frag._lazy.frame
    img = to_white_background(frag, colors)
    ref = C._oracle_frame(sc, R, T)
    same = C._check_frame("frame path 90x75 K=18", frag, img, ref, max_flips=3)
    g_img = np.random.default_rng(6).normal(size=ref["image"].shape) * same[..., None]
    (img * t(g_img)).sum().backward()
    C._check_grads("frame path", (colors.grad, gm.verts.grad, gm.sigmas.grad), C._oracle_grads(sc, ref, g_img), mult=1)


def test_camera_that_wants_a_gradient_takes_the_ray_bundle(hip_lib):
    """A pose under optimisation (R / T with requires_grad) is not a fixed camera: the renderer generates the bundle with the
    differentiable ray kernel as before, and the gradient reaches T."""
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    verts, sig, cols = _scene(500, seed=2)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 20.0)
    size = (48, 48)
    cams = PerspectiveCameras(focal_length=60.0, principal_point=((24.0, 24.0),), image_size=(size,), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=size, max_assign=10, max_point_per_bin=-1)).to(DEV)
    Tt = t(T, rg=True)
    frag = renderer(GaussianMeshes(t(verts), t(sig)).to(DEV), R=t(R), T=Tt)
    assert frag._lazy is None or not frag._lazy.frame
    to_white_background(frag, t(cols)).sum().backward()
    assert Tt.grad is not None and torch.isfinite(Tt.grad).all() and float(Tt.grad.abs().max()) > 0


@pytest.mark.parametrize("B,inverse_sigma", [(1, False), (2, True)])
def test_training_pattern_on_the_frame_path(hip_lib, B, inverse_sigma):
    """interpolate_attr + get_silhouette (demo/ShapeFitting.py:217-222) on the frame path's fragments: forward bits of the
    ray-bundle chain, and the one-launch merge backward (voge_frame_merge_bwd_iso) against the acc + fill + finish form."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, interpolate_attr
    from voge_amd.cameras import PerspectiveCameras
    verts, sig, cols = _scene(2500, seed=40 + B)
    if inverse_sigma:
        sig = (1.0 / sig).astype(np.float32)
    R, T = _views(B, seed=8)
    size = (72, 88)
    rng = np.random.default_rng(1)
    w_rgb, w_sil = t(rng.normal(size=(B,) + size + (3,))), t(rng.normal(size=(B,) + size))
    out = []
    for frame, direct in ((True, True), (True, False), (False, False)):
        cams = PerspectiveCameras(focal_length=100.0, principal_point=((44.0, 36.0),), image_size=(size,), device=DEV)
        renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=size, max_assign=15, max_point_per_bin=-1,
                                                                 inverse_sigma=inverse_sigma)).to(DEV)
        gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
        colors = t(cols, rg=True)
        old = (ops.FRAME_PATH, ops.FRAME_DIRECT_BWD)
        ops.FRAME_PATH, ops.FRAME_DIRECT_BWD = frame, direct
        try:
            frag = renderer(gm, R=t(R), T=t(T))
            assert (frag._lazy is not None and frag._lazy.frame) == frame
            rgb = interpolate_attr(frag, colors.repeat(B, 1) if B > 1 else colors)
            sil = get_silhouette(frag)
            ((rgb * w_rgb).sum() + (sil * w_sil).sum()).backward()
        finally:
            ops.FRAME_PATH, ops.FRAME_DIRECT_BWD = old
        out.append((rgb.detach(), sil.detach(), colors.grad, gm.verts.grad, gm.sigmas.grad))
    for other in out[1:]:
        assert torch.equal(out[0][0], other[0]) and torch.equal(out[0][1], other[1])
        for ga, gb in zip(out[0][2:], other[2:]):
            assert float((ga - gb).abs().max()) <= 2e-5 * max(1.0, float(gb.abs().max()))


@pytest.mark.parametrize("sig_kind", ["scalar", "diag", "full"])
def test_max_assign_above_the_lds_cap_is_refused_cleanly(hip_lib, sig_kind):
    """VOGE_MAX_K = 256 (include/voge_hip.h): the top-K lists of a tile live in LDS, where the reference keeps them in global
    memory without a cap (ray_trace_voge.cu:197-212; its demos stop at max_assign = 102).  A larger max_assign must come back
    as ONE clear error from every public route -- renderer (all three sigma forms), ray_tracing, aggregation -- before
    anything is launched, and leave the process able to render."""
    from voge_amd import _lib, scenes
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    aniso = {"scalar": False, "diag": "diag", "full": True}[sig_kind]
    verts, sig, cols = scenes.random_gaussians(400, seed=3, anisotropic=aniso, r_lo=0.05, r_hi=0.1)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 20.0)
    size = (40, 40)
    cams = PerspectiveCameras(focal_length=50.0, principal_point=((20.0, 20.0),), image_size=(size,), device=DEV)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    for bins in (-1, None):
        renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=size, max_assign=300, max_point_per_bin=bins)).to(DEV)
        with pytest.raises(_lib.VogeHipError, match="K exceeds VOGE_MAX_K"):
            renderer(gm, R=t(R), T=t(T))
    ok = GaussianRenderer(cams, GaussianRenderSettings(image_size=size, max_assign=256, max_point_per_bin=-1)).to(DEV)
    img = to_white_background(ok(gm, R=t(R), T=t(T)), t(cols))
    assert torch.isfinite(img).all() and float(img.min()) < 1.0
    if sig_kind == "scalar":
        from voge_amd import ops
        big = torch.zeros((1, 4, 4, 300), device=DEV)
        with pytest.raises(_lib.VogeHipError, match="K exceeds VOGE_MAX_K"):
            ops.composite(torch.zeros((1, 4, 4, 300), dtype=torch.int32, device=DEV), big, big, big)


def test_hit_length_of_the_frame_path_is_differentiable_on_demand(hip_lib):
    """The camera-input trace makes no autograd node; Fragments.vert_hit_length becomes a differentiable alias when it is read
    (ops._HitLength).  Its gradient -- alone and next to an image loss on the same fragments -- against the ray-bundle chain's."""
    from voge_amd import ops
    verts, sig, cols = _scene(1200, seed=77)
    R, T = _views(1, seed=5)
    out = []
    for frame in (True, False):
        frag, img, gm, colors, lz = _render(verts, sig, cols, R, T, (56, 64), 10, frame=frame)
        hl = frag.vert_hit_length
        assert hl.requires_grad and hl.data_ptr() == lz.sel_len.data_ptr()
        hit = (hl < 1e9).float()
        ((hl * hit).sum() * 0.01 + (img * img).sum()).backward()
        out.append((n(hl), gm.verts.grad, gm.sigmas.grad, colors.grad))
    assert np.array_equal(out[0][0], out[1][0])
    for ga, gb in zip(out[0][1:], out[1][1:]):
        assert float((ga - gb).abs().max()) <= 2e-5 * max(1.0, float(gb.abs().max()))
    # rendered without grad: the plain tensor
    with torch.no_grad():
        frag, img, gm, colors, lz = _render(verts, sig, cols, R, T, (56, 64), 10, frame=True, grad=False)
        assert not frag.vert_hit_length.requires_grad


def _general_scene(N, seed, kind, per_view=0):
    """(N,3) per-axis or (N,3,3) L L^T sigmas (scenes.random_gaussians' forms); per_view = B: a [B,N,...] stack of them."""
    from voge_amd import scenes
    verts, sig, cols = scenes.random_gaussians(N, seed=seed, anisotropic=("diag" if kind == 1 else True), r_lo=0.04, r_hi=0.09)
    if per_view:      # (a [B,N,...] stack of sigmas AND of vertices: nothing shared between the views)
        rng = np.random.default_rng(seed)
        sig = np.stack([sig * rng.uniform(0.8, 1.25) for _ in range(per_view)]).astype(np.float32)
        verts = np.stack([verts + rng.normal(size=verts.shape).astype(np.float32) * 0.01 for _ in range(per_view)]).astype(np.float32)
    return verts, sig, cols


@pytest.mark.parametrize("kind,B,per_view,route", [
    (1, 1, 0, "white"), (2, 1, 0, "white"), (1, 2, 0, "white"), (2, 2, 2, "white"), (1, 2, 2, "merge"),
    (1, 1, 0, "merge"), (2, 2, 0, "merge"), (1, 1, 0, "weights"), (2, 1, 0, "weights+white"), (1, 2, 0, "hit_length"),
])
def test_general_forms_on_the_frame_path(hip_lib, kind, B, per_view, route):
    """(N,3) / (N,3,3) sigmas through voge_frame_trace_fwd_gen + voge_frame_shade_fwd_rec + voge_frame_bwd_gen (the camera and the
    user's own arrays in, the user's own gradients out: no ray launch, no preamble launch either way, no pack, no fill) against the
    ray-bundle chain (general_preamble + voge_trace_lean_fwd + voge_fragment_*bwd): forward bits, gradients to the atomics' order,
    on every consumer route of the fragments."""
    from voge_amd import ops
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, interpolate_attr, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    verts, sig, cols = _general_scene(1500, seed=60 + kind + B, kind=kind, per_view=per_view)
    R, T = _views(B, seed=kind + 2 * B)
    size = (72, 80)
    rng = np.random.default_rng(3)
    w_img, w_sil, w_w = t(rng.normal(size=(B,) + size + (3,))), t(rng.normal(size=(B,) + size)), t(rng.normal(size=(B,) + size + (14,)))
    out = []
    for frame in (True, False):
        cams = PerspectiveCameras(focal_length=95.0, principal_point=((40.5, 35.0),), image_size=(size,), device=DEV)
        renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=size, max_assign=14, max_point_per_bin=-1)).to(DEV)
        gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
        colors = t(cols, rg=True)
        colsB = colors.repeat(B, 1) if B > 1 else colors
        old = ops.FRAME_PATH
        ops.FRAME_PATH = frame
        try:
            frag = renderer(gm, R=t(R), T=t(T))
            assert (frag._lazy is not None and frag._lazy.frame and frag._lazy.gen is not None) == frame
            if route == "white":
                vals = [to_white_background(frag, colsB)]
                loss = (vals[0] * w_img).sum()
            elif route == "merge":
                vals = [interpolate_attr(frag, colsB), get_silhouette(frag)]
                loss = (vals[0] * w_img).sum() + (vals[1] * w_sil).sum()
            elif route == "weights":
                vals = [frag.vert_weight]
                loss = (vals[0] * w_w).sum()
            elif route == "weights+white":      # a second consumer of the weights next to the image
                img = to_white_background(frag, colsB)
                vals = [img, frag.vert_weight]
                loss = (img * w_img).sum() + (frag.vert_weight * w_w).sum()
            else:
                hl = frag.vert_hit_length
                vals = [hl, to_white_background(frag, colsB)]
                loss = (torch.where(hl < 1e9, hl, torch.zeros_like(hl)) * w_w).sum() * 0.01 + (vals[1] * w_img).sum()
            loss.backward()
        finally:
            ops.FRAME_PATH = old
        out.append(([v.detach() for v in vals], [frag.vert_index, frag.valid_num], [gm.verts.grad, gm.sigmas.grad, colors.grad]))
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert torch.equal(a, b), route
    for name, ga, gb in zip(("verts", "sigmas", "colors"), out[0][2], out[1][2]):
        if gb is None:
            assert ga is None or float(ga.abs().max()) == 0.0, (route, name)
            continue
        assert ga is not None and ga.shape == gb.shape, (route, name)
        assert float((ga - gb).abs().max()) <= 2e-5 * max(1.0, float(gb.abs().max())), (route, name)
