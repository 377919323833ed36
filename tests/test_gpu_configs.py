"""GPU parity at BASELINE.json's five configs, at their full sizes, plus the overflow fallbacks of the
binning levels.  (VERDICT r1 "Next round" item 1.)

cfg1 866 cuboid Gaussians / 256^2 / K=20   -> tests/test_gpu_parity.py::test_whole_frame_config1_vs_oracle
cfg2 8171 bunny Gaussians / 256^2 / K=40   -> test_config2_bunny_fwd_bwd           (fwd + bwd vs the oracle chain)
cfg3 50k / 512^2 / K=40                    -> test_gpu_parity.py::test_full_size_properties_config3 + test_config3_band_gradients
cfg4 200k / 1024^2 / K=40                  -> test_config4_full_size                (properties, oracle windows, determinism,
                                                                                      gradients of a row band vs the oracle)
cfg5 2562 / 128^2 / K=25 (ShapeFitting)    -> test_config5_shapefit_frame_fwd_bwd  (one 5-view fwd+bwd frame vs the oracle)

The oracle is fp64 brute force; where a whole frame would take minutes it is cropped to windows / a band of rows
(bands are bit-identical to the whole frame: test_row_bands_equal_whole_frame_and_default_bins).
"""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import camera_np
from util import TOL, bunny_scene, compare_trace, log_line, random_scene, _report_flips

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a, dtype=torch.float32, rg=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device=DEV, requires_grad=rg)


def n(x):
    return x.detach().cpu().numpy()


def _scene_cfg(name):
    from voge_amd import scenes
    N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
    verts, sig, cols = scenes.random_gaussians(N, seed=0)
    return dict(verts=verts, sigmas=sig, colors=cols, focal=focal, principal=pp, image_size=(H, W), dist=dd, elev=el,
                azim=az, K=K)


def _render(sc, rows=None, grad=True, max_point_per_bin=-1, views=None):
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    size = sc["image_size"]
    if views is None:
        R, T = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
    else:
        R, T = views
    cams = PerspectiveCameras(focal_length=sc["focal"], principal_point=(sc["principal"],), image_size=(size,), device=DEV)
    st = GaussianRenderSettings(image_size=size, max_assign=sc["K"], max_point_per_bin=max_point_per_bin)
    renderer = GaussianRenderer(cams, st).to(DEV)
    gm = GaussianMeshes(t(sc["verts"]), t(sc["sigmas"])).to(DEV)
    colors = t(sc["colors"], rg=grad)
    kw = {} if rows is None else dict(rows=rows)
    frag = renderer(gm, R=t(R), T=t(T), **kw)
    B = np.asarray(R).reshape(-1, 3, 3).shape[0]
    # (a batch of B views addresses attribute rows b * N + n: the attributes are tiled over the batch, Aggregation.py:120)
    img = to_white_background(frag, colors.repeat(B, 1) if B > 1 else colors)
    return frag, img, gm, colors, (R, T)


def _oracle_frame(sc, R, T, rows=None, cols=None):
    """The oracle's forward chain on the frame (or the row band / pixel window) of scene `sc`."""
    H, W = sc["image_size"]
    rays, origin = camera_np.pixel_rays(R, T, sc["focal"], sc["principal"], (H, W))
    B = rays.shape[0]
    if rows is not None:
        rays = rays[:, rows[0]:rows[1]]
    if cols is not None:
        rays = rays[:, :, cols[0]:cols[1]]
    rays = np.ascontiguousarray(rays)
    verts = np.asarray(sc["verts"], np.float32)
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(np.asarray(sc["sigmas"], np.float32))).astype(np.float32)
    isg = np.ascontiguousarray(np.broadcast_to(isg[None], (B,) + isg.shape))
    thr_act = oracle.thr_act_of(0.01)
    idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, sc["K"], thr_act)
    w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
    colsB = np.tile(np.asarray(sc["colors"]), (B, 1))
    rgb = oracle.merge_fwd(colsB, idx, w, vn)
    img, sil = oracle.blend_fwd(rgb, w)
    return dict(rays=rays, mus=mus, isg=isg, idx=idx, len=ln, act=act, dsd=dsd, weight=w, valid_num=vn, rgb=rgb,
                image=img, silhouette=sil, thr_act=thr_act, colsB=colsB)


def _oracle_grads(sc, ref, g_img):
    """Backward chain of the oracle for loss = sum(img * g_img): grads of colours [N,3], verts [N,3] and of the
    user's sigmas ([N] scalars: d/ds of A = 2 s I; [N,3,3]: d/dS of A = 2 S)."""
    x = ref["rgb"] + (1 - ref["silhouette"])[..., None]
    g_rgb = g_img * (x < 1)
    g_sil = -(g_rgb.sum(-1)) * (ref["weight"].sum(-1) < 1)
    g_attr, g_w = oracle.merge_bwd(ref["colsB"], ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w + g_sil[..., None], 1.0)
    _, g_mu, g_A = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
    B = ref["rays"].shape[0]
    N = np.asarray(sc["verts"]).shape[0]
    g_attr = g_attr.reshape(B, N, -1).sum(0)
    g_mu = g_mu.reshape(B, N, 3).sum(0)
    g_A = g_A.reshape(B, N, 3, 3).sum(0)
    nd = np.asarray(sc["sigmas"]).ndim      # [N]: A = 2 s I; [N,3]: A = 2 diag(s); [N,3,3]: A = 2 S
    g_sig = {1: 2 * np.einsum("nii->n", g_A), 2: 2 * np.einsum("nii->ni", g_A), 3: 2 * g_A}[nd]
    return g_attr, g_mu, g_sig


def _check_frame(label, frag, img, ref, max_flips, img_tol=TOL):
    """Forward fragments / image against the oracle; returns the mask of pixels with identical index lists."""
    idx = n(frag.vert_index)
    same = (idx == np.where(ref["idx"] < 0, 0, ref["idx"])).all(-1) | (idx == ref["idx"]).all(-1)
    _report_flips(label, (~same).sum(), same.size)
    assert (~same).sum() <= max_flips, f"{label}: {(~same).sum()} of {same.size} pixels flipped (ceiling {max_flips})"
    assert (n(frag.valid_num)[same] == ref["valid_num"][same]).all()
    assert np.abs(n(frag.vert_weight)[same] - ref["weight"][same]).max(initial=0.0) < TOL
    hit = (ref["idx"] >= 0) & same[..., None]
    err = np.abs(n(frag.vert_hit_length)[hit] - ref["len"][hit]) / np.maximum(1.0, np.abs(ref["len"][hit]))
    assert err.max(initial=0.0) < TOL
    assert np.abs(n(img)[same] - ref["image"][same]).max(initial=0.0) < img_tol
    # a flipped member moves a weight by <= thr e^0.5 -- unless the list is full: then the flip also swaps the LAST member for
    # another Gaussian of any weight (tools/soak.py, seed 201 case 77: K = 8, a member at the activation threshold with
    # weight 0.014 missing, a far one with weight 0.50 in its place), so only lists with room are held to the bound
    roomy = same | (ref["valid_num"] < idx.shape[-1])
    assert np.abs(n(img) - ref["image"])[roomy].max(initial=0.0) < 0.05
    return same


def _check_grads(label, got, want, mult):
    out = {}
    for name, g, w in zip(("colors", "verts", "sigmas"), got, want):
        g = n(g).astype(np.float64).reshape(w.shape)
        scale = max(1.0, np.abs(w).max())
        err = np.abs(g - w).max()
        out[name] = err / scale
        assert err <= mult * TOL * scale, f"{label} {name}: {err:.3e} vs scale {scale:.3e} (allowed {mult} x {TOL})"
    line = f"[parity] {label} gradient errors / scale: " + ", ".join(f"{k} {v:.2e}" for k, v in out.items()) + f" (allowed {mult * TOL:.1e})"
    print(line)
    import os
    log = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(log):
        with open(os.path.join(log, "parity_flips.txt"), "a") as f:
            f.write(line + "\n")
    return out


# ----------------------------------------------------------------------------------------------- cfg2
def test_config2_bunny_fwd_bwd(hip_lib):
    """BASELINE config 2 ("fwd+bwd vs reference CUDA numerics"): the bunny's 8171 Gaussians have A ~ 2e4..1e6 at
    distance 6 -- the ill-conditioned case.  Whole frame forward and the gradients of a random image loss against
    the fp64 oracle chain."""
    sc = bunny_scene()
    frag, img, gm, colors, (R, T) = _render(sc)
    ref = _oracle_frame(sc, R, T)
    same = _check_frame("cfg2 bunny 256^2 K=40", frag, img, ref, max_flips=100)
    # The loss leaves out the pixels whose member set differs (a candidate within rounding of the act threshold or
    # of the K-th depth: a discrete difference, counted and bounded above): what remains is pure arithmetic.
    g_img = np.random.default_rng(2).normal(size=ref["image"].shape) * same[..., None]
    (img * t(g_img)).sum().backward()
    want = _oracle_grads(sc, ref, g_img)
    # Round 4 (tools/cfg2_grad_gap.py, profiles/r4_cfg2_grad_gap.txt): the 1.9e-4 of rounds 1-3 came from the TRACE backward --
    # v = mu - t d keeps t's fp32 rounding along d (2e-4 of |v| at |mu| ~ 6, |v| ~ 1e-3) and g_mu scales it by 2 a g_act
    # ~ 1e6; one projection of v off d took that stage from 2.8e-4 to 1.4e-7 of scale.  What is left (1.5e-4) is the fp32
    # VALUE of len itself (2-5 ulp at 6 = 1.2e-6) seen through s = sqrt(dsd) ~ 1e3 in the composite's erf arguments: the
    # oracle's fp64 backward chain fed with the fp32 forward values alone shows 1.4e-4.  Any fp32 path -- the reference's
    # included, which stores len as float -- carries it; the reference's own fp32 operation order does far worse (below).
    got = _check_grads("cfg2", (colors.grad, gm.verts.grad, gm.sigmas.grad), want, mult=2)      # measured 1.5e-4 (verts)
    # Round 5, the FORMAT's floor: the oracle's fp64 chain -- composite, merge, blend and their backward,
    # trace backward -- with nothing changed but len rounded to the NEAREST fp32 (<= 0.5 ulp; act / dsd left in fp64).  On
    # this frame that alone moves the vertex gradient by 1.46e-4 of scale: above north_star's 1e-4, i.e. no path that hands
    # len over as a float -- the reference's `float` tensors included -- can meet 1e-4 against an fp64 truth here, however the
    # float was computed (the HIP len's 2.5 ulp cost 0.05e-4 more).  The 2 x tolerance above stands on this number; HIP is
    # held to within 10 % of it.
    fl = dict(ref)
    fl["len"] = ref["len"].astype(np.float32).astype(np.float64)
    fl["weight"], fl["valid_num"] = oracle.composite_fwd(ref["idx"], ref["act"], fl["len"], ref["dsd"], 1.0)
    fl["rgb"] = oracle.merge_fwd(ref["colsB"], ref["idx"], fl["weight"], fl["valid_num"])
    _, fl["silhouette"] = oracle.blend_fwd(fl["rgb"], fl["weight"])
    floor = np.abs(_oracle_grads(sc, fl, g_img)[1] - want[1]).max() / max(1.0, np.abs(want[1]).max())
    from util import log_line
    log_line(f"[parity] cfg2 verts gradient: fp64 chain with len rounded to nearest fp32 (the format's floor) {floor:.3e} of scale; "
             f"HIP {got['verts']:.3e} = {got['verts'] / floor:.3f} x the floor")
    assert floor > TOL, "the fp32-len floor fell below north_star's tolerance: tighten _check_grads('cfg2') to mult=1"
    assert got["verts"] <= 1.10 * floor
    # the reference's arithmetic floor on the same frame: the fp32 reference-order oracle (act = mu^T A mu - (mu^T A d)^2/dsd
    # cancels catastrophically here), same lists, same loss
    f32 = dict(ref)
    i32, l32, a32, d32 = oracle.trace_fwd(ref["mus"], ref["isg"], ref["rays"], sc["K"], ref["thr_act"], precision="f32")
    agree = (i32 == ref["idx"]).all(-1)[..., None]
    f32["len"], f32["act"], f32["dsd"] = (np.where(agree, x, ref[k]) for x, k in ((l32, "len"), (a32, "act"), (d32, "dsd")))
    x = ref["rgb"] + (1 - ref["silhouette"])[..., None]
    g_rgb = g_img * (x < 1)
    _, g_w = oracle.merge_bwd(ref["colsB"], ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
    g_w = g_w - ((g_rgb.sum(-1)) * (ref["weight"].sum(-1) < 1))[..., None]
    ga, gl, gd = oracle.composite_bwd(f32["act"], f32["len"], f32["dsd"], g_w, 1.0, precision="f32")
    _, g_mu32, _ = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], gl, ga, gd, precision="f32")
    err32 = np.abs(np.asarray(g_mu32, np.float64).reshape(want[1].shape) - want[1]).max() / max(1.0, np.abs(want[1]).max())
    log_line(f"[parity] cfg2 verts gradient: HIP {got['verts']:.2e} of scale; the fp32 reference-order oracle chain {err32:.2e}")
    assert got["verts"] <= err32


# ----------------------------------------------------------------------------------------------- cfg3
def test_config3_band_gradients(hip_lib):
    """cfg3 (50k / 512^2 / K=40): forward + gradients of an 8-row band through the renderer (rows=) against the
    oracle on the same band -- the band runs the same coarse / super-tile / tile binning as the frame."""
    sc = _scene_cfg("cfg3_50k_512")
    rows = (252, 260)
    frag, img, gm, colors, (R, T) = _render(sc, rows=rows)
    ref = _oracle_frame(sc, R, T, rows=rows)
    same = _check_frame("cfg3 rows 252..259", frag, img, ref, max_flips=4)
    g_img = np.random.default_rng(3).normal(size=ref["image"].shape) * same[..., None]
    (img * t(g_img)).sum().backward()
    _check_grads("cfg3 band", (colors.grad, gm.verts.grad, gm.sigmas.grad), _oracle_grads(sc, ref, g_img), mult=0.6)      # measured 2.6e-5


def _big_host():
    """The fp64 brute-force oracle is OpenMP code: a whole cfg3 frame (50k x 262 144 pairs, forward + backward) is ~8 s on the
    GPU box's 256 host threads and minutes on the 8-core build container.  The whole-frame / 64-row tests run where the
    host has >= 64 threads (VOGE_WHOLE_FRAME_PARITY=1 forces them, =0 skips them); the 8- / 4-row bands above stay as the
    small-host form of the same check."""
    force = os.environ.get("VOGE_WHOLE_FRAME_PARITY")
    if force is not None:
        return force != "0"
    return (os.cpu_count() or 1) >= 64


def _classify_flips(label, frag, ref, same, ceilings):
    """Flipped pixels by kind: a different member SET (returned), or the same members with depth-tied ones swapped (every
    moved member is checked to tie the depth of the slot it moved to within the fp32 rounding of len)."""
    gi, ri, rl = n(frag.vert_index)[~same], ref["idx"][~same], ref["len"][~same]
    set_flips = 0
    for g, r, l in zip(gi, ri, rl):
        gs, rs = set(g[g > 0].tolist()), set(r[r > 0].tolist())      # (slot value 0 is merge_final's rewritten -1 or Gaussian 0)
        if gs != rs:
            set_flips += 1
            continue
        pos = {int(v): k for k, v in enumerate(r) if v >= 0}
        for k, v in enumerate(g[:len(pos)]):
            if int(v) in pos and pos[int(v)] != k:      # moved: its depth must tie the depth of the slot it moved to
                assert abs(l[pos[int(v)]] - l[k]) <= 2e-6 * max(1.0, abs(l[k])), "members swapped away from a depth tie"
    log_line(f"[parity] {label}: {int((~same).sum())} flipped pixels of {same.size} = {set_flips} with a different member set "
             f"+ {int((~same).sum()) - set_flips} order swaps of depth ties (ceilings {ceilings[0]} / {ceilings[1]} in all)")
    return set_flips


def test_config3_whole_frame_vs_oracle(hip_lib):
    """the HEADLINE config's whole 512^2 frame -- fragments, image and all three gradients of a random
    image loss -- against the fp64 oracle chain (ray_trace_voge.cu:135-217, Aggregation.py:82-107 restated).  Ceilings:
    flipped index lists <= 0.01 % of the pixels, every gradient within 1e-4 of its scale."""
    if not _big_host():
        pytest.skip("whole-frame oracle parity needs a host with >= 64 threads (the 8-row band test covers small hosts)")
    sc = _scene_cfg("cfg3_50k_512")
    H, W = sc["image_size"]
    frag, img, gm, colors, (R, T) = _render(sc)
    ref = _oracle_frame(sc, R, T)
    # A "flip" is a pixel whose ORDERED index list differs.  Two kinds: (1) the same members with two of them swapped -- their
    # depths lie within the fp32 rounding of len (a few ulp of ~3.5), the order any fp32 implementation, the reference's
    # included, is free to differ on against an fp64 truth: expected count = lit pixels x adjacent pairs x density x 2 eps
    # ~ 160 000 x 17 x 18 x 6e-7 ~ 30; (2) a different member SET (a candidate within rounding of the activation threshold or
    # of the K-th depth).  VERDICT r5's ceiling of 0.01 % is held on the second kind; all flips together stay under 0.02 %,
    # and every swap is checked to be a near-tie.
    same = _check_frame("cfg3 whole frame", frag, img, ref, max_flips=(H * W) // 5000)
    set_flips = _classify_flips("cfg3 whole frame", frag, ref, same, (same.size // 10000, same.size // 5000))
    assert set_flips <= (H * W) // 10000
    g_img = np.random.default_rng(33).normal(size=ref["image"].shape) * same[..., None]
    (img * t(g_img)).sum().backward()
    _check_grads("cfg3 whole frame", (colors.grad, gm.verts.grad, gm.sigmas.grad), _oracle_grads(sc, ref, g_img), mult=1)


def test_config4_64_rows_vs_oracle(hip_lib):
    """cfg4 (200k / 1024^2): 64 rows (two rows of super-tiles, 65 536 pixels -- the pair count of a whole cfg3 frame)
    through the renderer's rows= against the oracle, forward and gradients."""
    if not _big_host():
        pytest.skip("64-row oracle parity at cfg4 needs a host with >= 64 threads (test_config4_full_size covers small hosts)")
    sc = _scene_cfg("cfg4_200k_1024")
    rows = (480, 544)
    frag, img, gm, colors, (R, T) = _render(sc, rows=rows)
    ref = _oracle_frame(sc, R, T, rows=rows)
    # (200k Gaussians: four times cfg3's depth density and every lit list full, so ties at the K-th depth and between
    #  neighbours are that much more frequent -- measured 24 flips of 65 536; ceilings at 1.7 x that)
    same = _check_frame("cfg4 rows 480..543", frag, img, ref, max_flips=40)
    assert _classify_flips("cfg4 rows 480..543", frag, ref, same, (12, 40)) <= 12      # (measured: 5 member-set flips + 19 tie swaps)
    g_img = np.random.default_rng(44).normal(size=ref["image"].shape) * same[..., None]
    (img * t(g_img)).sum().backward()
    _check_grads("cfg4 64 rows", (colors.grad, gm.verts.grad, gm.sigmas.grad), _oracle_grads(sc, ref, g_img), mult=1)


# ----------------------------------------------------------------------------------------------- cfg4
def test_config4_full_size(hip_lib):
    """BASELINE config 4 (200k Gaussians, 1024x1024, K=40) at full size: the coarse bin0 level on 64 regions,
    16384 sweep tiles (the tile_order kernel's chunk limit), offsets 4x those of cfg3.
    Size-independent properties on the whole frame, determinism, fragments in five 24x24 windows against the
    brute-force oracle, and the gradients of a 4-row band against the oracle chain."""
    sc = _scene_cfg("cfg4_200k_1024")
    H, W = sc["image_size"]
    K, N = sc["K"], sc["verts"].shape[0]
    frag, img, gm, colors, (R, T) = _render(sc)
    idx, w, vn, hl = n(frag.vert_index), n(frag.vert_weight), n(frag.valid_num), n(frag.vert_hit_length)
    slot = np.arange(K)[None, None, None]
    filled = slot < vn[..., None]
    assert np.isfinite(w).all() and (w >= 0).all() and (w[~filled] == 0).all()
    assert (hl[~filled] == np.float32(1e10)).all()
    assert (np.diff(hl, axis=-1)[filled[..., 1:]] >= 0).all(), "hit lengths must ascend within the valid prefix"
    assert (idx[filled] >= 0).all() and (idx[filled] < N).all()
    srt = np.sort(np.where(filled, idx, -1 - slot), axis=-1)
    assert (np.diff(srt, axis=-1) != 0).all(), "a Gaussian may appear once per pixel"
    im = n(img)
    assert im.min() >= 0 and im.max() <= 1 and vn.max() == K and (vn == 0).mean() > 0.05
    # determinism of the forward (two more renders)
    frag_b, img_b, *_ = _render(sc, grad=False)
    assert torch.equal(img_b, img.detach()) and torch.equal(frag_b.vert_weight, frag.vert_weight.detach())
    assert torch.equal(frag_b.vert_hit_length, frag.vert_hit_length.detach())
    # oracle windows (200k x 576 pairs each)
    S = 24
    flips = 0
    for y0, x0 in ((500, 500), (0, 0), (H - S, W - S), (300, 700), (777, 123)):
        ref = _oracle_frame(sc, R, T, rows=(y0, y0 + S), cols=(x0, x0 + S))
        sub = type(frag)(**{k: getattr(frag, k)[:, y0:y0 + S, x0:x0 + S] for k in frag._fields})
        same = _check_frame(f"cfg4 window ({y0},{x0})", sub, img[:, y0:y0 + S, x0:x0 + S], ref, max_flips=4)
        flips += int((~same).sum())
    assert flips <= 8
    # backward of the whole frame is finite; gradients of a 4-row band against the oracle
    img.sum().backward()
    for g in (gm.verts.grad, gm.sigmas.grad, colors.grad):
        assert torch.isfinite(g).all() and g.abs().max() > 0
    rows = (510, 514)
    frag2, img2, gm2, colors2, _ = _render(sc, rows=rows)
    assert torch.equal(img2.detach(), img.detach()[:, rows[0]:rows[1]])      # the band IS the frame's rows
    ref = _oracle_frame(sc, R, T, rows=rows)
    same = _check_frame("cfg4 rows 510..513", frag2, img2, ref, max_flips=4)
    g_img = np.random.default_rng(4).normal(size=ref["image"].shape) * same[..., None]
    (img2 * t(g_img)).sum().backward()
    _check_grads("cfg4 band", (colors2.grad, gm2.verts.grad, gm2.sigmas.grad), _oracle_grads(sc, ref, g_img), mult=0.3)      # measured 1.5e-5


# ----------------------------------------------------------------------------------------------- cfg5
def test_config5_shapefit_frame_fwd_bwd(hip_lib):
    """BASELINE config 5 at its real size (demo/ShapeFitting.py:214-296: ico-sphere level 4 = 2562 Gaussians,
    128x128, max_assign 25, 5 views per iteration, max_point_per_bin=-1): ONE iteration's forward + backward --
    silhouette and rgb losses on five views of one shared Gaussian set -- against the full oracle chain."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("shape_fitting_demo", os.path.join(root, "demo", "ShapeFitting.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    v, _ = demo.ico_sphere(4)
    assert v.shape == (2562, 3)
    rng = np.random.default_rng(5)
    verts = (np.asarray(v, np.float32) * (1.0 + 0.05 * rng.normal(size=(2562, 1)))).astype(np.float32)
    # ShapeFitting.py's sigma rule: one scalar per vertex from the mean edge length (converter percentage 0.5..0.6)
    sig = np.full(2562, 1.0 / (0.05 ** 2 / (2 * np.log(1 / 0.6))), np.float32) * rng.uniform(0.8, 1.25, 2562).astype(np.float32)
    cols = rng.uniform(0, 1, (2562, 3)).astype(np.float32)
    sc = dict(verts=verts, sigmas=sig, colors=cols, focal=126.0, principal=(64.0, 64.0), image_size=(128, 128), K=25)
    R5, T5 = camera_np.look_at_view_transform([2.7] * 5, [0.0, 30.0, -20.0, 10.0, 45.0], [-180.0, -120.0, -40.0, 60.0, 160.0])
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, to_white_background
    from voge_amd.cameras import PerspectiveCameras
    cams = PerspectiveCameras(focal_length=126.0, principal_point=((64.0, 64.0),), image_size=((128, 128),), device=DEV)
    renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(128, 128), max_assign=25, max_point_per_bin=-1)).to(DEV)
    gm = GaussianMeshes(t(verts), t(sig)).to(DEV)
    colors = t(cols, rg=True)
    want = [np.zeros((2562, 3)), np.zeros((2562, 3)), np.zeros(2562)]
    flips = 0
    loss = 0.0
    for j in range(5):      # the reference renders the views of an iteration one at a time (ShapeFitting.py:258-259)
        frag = renderer(gm, R=t(R5[j:j + 1]), T=t(T5[j:j + 1]))
        img = to_white_background(frag, colors)
        sil = get_silhouette(frag)
        ref = _oracle_frame(sc, R5[j:j + 1], T5[j:j + 1])
        same = _check_frame(f"cfg5 view {j} 128^2 K=25", frag, img, ref, max_flips=10)
        flips += int((~same).sum())
        assert (ref["valid_num"] > 0).mean() > 0.3
        # the demo's loss: mean squared error of rgb and of the silhouette against targets (ShapeFitting.py:262-271)
        tgt_rgb = rng.uniform(0, 1, ref["image"].shape)
        tgt_sil = (rng.uniform(0, 1, ref["silhouette"].shape) > 0.5).astype(np.float64)
        keep = t(same.astype(np.float32))          # pixels with a different member set leave the loss (see cfg2)
        loss = loss + ((((img - t(tgt_rgb)) ** 2) * keep[..., None]).sum() / tgt_rgb.size
                       + (((sil - t(tgt_sil)) ** 2) * keep).sum() / tgt_sil.size) / 5
        # oracle: d loss / d img and d loss / d sil pushed through the same chain
        g_img = 2 * (ref["image"] - tgt_rgb) / tgt_rgb.size / 5 * same[..., None]
        g_silh = 2 * (ref["silhouette"] - tgt_sil) / tgt_sil.size / 5 * same
        x = ref["rgb"] + (1 - ref["silhouette"])[..., None]
        g_rgb = g_img * (x < 1)
        under = ref["weight"].sum(-1) < 1
        g_sumw = (-(g_rgb.sum(-1)) + g_silh) * under
        g_attr, g_w = oracle.merge_bwd(ref["colsB"], ref["idx"], ref["weight"], ref["valid_num"], g_rgb)
        g_act, g_len, g_dsd = oracle.composite_bwd(ref["act"], ref["len"], ref["dsd"], g_w + g_sumw[..., None], 1.0)
        _, g_mu, g_A = oracle.trace_bwd(ref["mus"], ref["isg"], ref["rays"], ref["idx"], g_len, g_act, g_dsd)
        want[0] += g_attr
        want[1] += g_mu
        want[2] += 2 * np.einsum("nii->n", g_A)
    assert flips <= 12
    loss.backward()
    # the mean-reduced loss makes the gradients ~1e-5: judge them at their own scale (relative to the largest entry)
    for name, g, wv in zip(("colors", "verts", "sigmas"), (colors.grad, gm.verts.grad, gm.sigmas.grad), want):
        err = np.abs(n(g).astype(np.float64) - wv).max()
        print(f"[parity] cfg5 {name}: max err {err:.3e}, largest entry {np.abs(wv).max():.3e}")
        log_line(f"[parity] cfg5 {name} gradient: {err / np.abs(wv).max():.2e} of the largest entry (allowed {TOL:.0e})")
        assert err <= TOL * np.abs(wv).max(), f"cfg5 {name}: {err:.3e} vs largest entry {np.abs(wv).max():.3e}"


# ----------------------------------------------------------------------------------- overflow fallbacks
def _wide_scene(N, seed):
    """Gaussians so wide that every one of them can touch every pixel (reach ~ the whole cube) with
    thr_activation = 0 (thr_act = 23): the super-tile and tile lists cannot cull anything."""
    rng = np.random.default_rng(seed)
    verts = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    r = rng.uniform(0.5, 0.9, N)
    sig = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
    return verts, sig


@pytest.mark.parametrize("N,H,W,K,what", [
    (5000, 64, 64, 10, "tile"),        # super-tile lists hold 5000 <= kBinCap entries, tile lists > kTileCap = 2048
    (10000, 64, 64, 10, "bin"),        # > kBinCap = 8192 survivors per super-tile: bin_count = -1, full stream
    (20000, 64, 96, 7, "bin+coarse"),  # the same behind the coarse region level (N >= 16384, >= 4 super-tiles)
])
def test_bin_and_tile_list_overflow_fallbacks(hip_lib, N, H, W, K, what):
    """kBinCap (trace_fwd.hip bin_kernel: `total > kBinCap -> bin_count = -1`) and kTileCap (bin2_kernel:
    `total > kTileCap -> tl_count = -1`) switch the sweep to the parent stream.  Wide Gaussians with
    thr_activation = 0 overflow them; results must still equal the brute-force oracle (every entry point)."""
    from voge_amd import ops
    verts, sig = _wide_scene(N, seed=N)
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 20.0)
    rays, origin = camera_np.pixel_rays(R, T, 70.0, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(0.0)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    assert (ref[0][..., -1] >= 0).all()                                   # every pixel's list is full
    # candidates per pixel: (nearly) all N pass act < thr_act
    a = np.ascontiguousarray(isg[..., 0, 0])
    got_iso = [n(x) for x in ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)]
    compare_trace(got_iso, ref, thr_act, min_match=0.995, label=f"overflow {what} iso N={N}")
    got_gen = [n(x) for x in ops.ray_trace_fine(t(mus).reshape(-1, 3), t(isg).reshape(-1, 3, 3), t(rays), None, thr_act, 10, K)]
    compare_trace(got_gen, ref, thr_act, min_match=0.995, label=f"overflow {what} general N={N}")
    for x, y in zip(got_iso, got_gen):
        assert np.array_equal(x, y)                                       # the two entry points agree bit for bit


@pytest.mark.parametrize("N,extent,what", [
    (9000, 0.22, "interleaved dealing (N < 16384): 563 Gaussians per slice, all inside one super-tile"),
    (40000, 0.30, "chunked dealing: 2-3 chunks of 1024 per slice, most of them inside four super-tiles"),
    (2562, 0.5, "ShapeFitting's size: no segment may overflow at all (160 Gaussians per slice)")])
def test_small_object_segment_overflow_is_handled_per_slice(hip_lib, N, extent, what):
    """A small object far away puts more than kSegCap = 512 of a slice's Gaussians into ONE super-tile.  Round 2 sent
    every tile of such a super-tile to the stream-everything fallback (the fitted ShapeFitting sphere: sweep 80 -> 780 us);
    now small sets are dealt Gaussian by Gaussian over the slices (trace_bin.h: deal_interleaved), and a segment that still
    overflows is re-tested from the per-Gaussian records inside binB.  Results must equal the brute-force oracle
    (scalar-sigma and general entry points, which share binB)."""
    from voge_amd import ops
    rng = np.random.default_rng(N)
    verts = rng.uniform(-extent, extent, (N, 3)).astype(np.float32)
    r = rng.uniform(0.006, 0.012, N)
    sig = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
    H = W = 96
    K = 12
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 20.0)
    rays, origin = camera_np.pixel_rays(R, T, 110.0, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(0.01)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    assert (ref[0][..., 0] >= 0).mean() > 0.02 and (ref[0] >= 0).sum() > 2000, what      # the object is hit
    a = np.ascontiguousarray(isg[..., 0, 0])
    got_iso = [n(x) for x in ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)]
    compare_trace(got_iso, ref, thr_act, min_match=0.998, label=f"small object N={N} iso")
    got_gen = [n(x) for x in ops.ray_trace_fine(t(mus).reshape(-1, 3), t(isg).reshape(-1, 3, 3), t(rays), None, thr_act, 10, K)]
    for x, y in zip(got_iso, got_gen):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("N,extent,r_lo,r_hi,K,expect", [
    (40000, 0.30, 0.006, 0.012, 12, "pooled"),      # 40 000 Gaussians behind ~22 pixels: every quad of the object is long
    (60000, 0.25, 0.004, 0.008, 40, "pooled"),      # denser still, K = 40 (full lists in the middle, short ones at the rim)
    (32000, 0.5, 0.6, 0.7, 8, "exhausted")])        # huge footprints: every Gaussian in every quad -> the pool runs out
def test_quads_with_more_candidates_than_the_lds_sort_use_pooled_lists(hip_lib, N, extent, r_lo, r_hi, K, expect):
    """A quad (16x16 pixels) with more than kQCap = 3008 candidates used to send its four tiles to the stream-everything
    fallback (trace 0.1 -> 3 ms at 50k Gaussians seen from 4x farther than cfg3).  binB now counts and
    scatters such a quad's candidates bucket by bucket into per-tile lists taken from a pool (trace_bin.h: long path);
    only an exhausted pool still falls back.  Either way the result equals the brute-force oracle."""
    from voge_amd import ops
    rng = np.random.default_rng(N)
    verts = rng.uniform(-extent, extent, (N, 3)).astype(np.float32)
    r = rng.uniform(r_lo, r_hi, N)
    sig = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
    H = W = 96
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 20.0)
    rays, origin = camera_np.pixel_rays(R, T, 110.0, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(0.01)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    assert (ref[0][..., 0] >= 0).mean() > 0.02
    a = np.ascontiguousarray(isg[..., 0, 0])
    got_iso = [n(x) for x in ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)]
    used, cap = ops.trace_pool_usage("cuda:0", 1, N, H, W)
    assert used > 0, "no quad took the long path: the scene does not test it"
    assert (used <= cap) == (expect == "pooled"), (used, cap)
    compare_trace(got_iso, ref, thr_act, min_match=0.998, label=f"long quads N={N} iso")
    got_gen = [n(x) for x in ops.ray_trace_fine(t(mus).reshape(-1, 3), t(isg).reshape(-1, 3, 3), t(rays), None, thr_act, 10, K)]
    for x, y in zip(got_iso, got_gen):
        assert np.array_equal(x, y)


def test_long_quads_and_segment_extensions_in_a_batch_of_views(hip_lib):
    """The pooled lists and binA's segment extensions with B = 2: two views of the same small dense object (their arenas
    and pool entries are per batch element / shared), against the brute-force oracle."""
    from voge_amd import ops
    N, K, H, W = 24000, 16, 96, 96
    rng = np.random.default_rng(7)
    verts = rng.uniform(-0.28, 0.28, (N, 3)).astype(np.float32)
    r = rng.uniform(0.006, 0.012, N)
    sig = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.0, 3.4], [10.0, -20.0], [20.0, 140.0])
    rays, origin = camera_np.pixel_rays(R, T, 110.0, (W / 2.0, H / 2.0), (H, W))
    assert rays.shape[0] == 2
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = np.ascontiguousarray(np.broadcast_to((2 * camera_np.expand_sigma(sig)).astype(np.float32)[None], (2, N, 3, 3)))
    thr_act = oracle.thr_act_of(0.01)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    a = np.ascontiguousarray(isg[..., 0, 0])
    got = [n(x) for x in ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)]
    used, cap = ops.trace_pool_usage("cuda:0", 2, N, H, W)
    assert 0 < used <= cap, (used, cap)
    compare_trace(got, ref, thr_act, min_match=0.998, label="long quads, two views")


# ------------------------------------------------------------------ round 4: the rebuilt scalar-sigma sweep (sweep_iso.h)
def _ab_library():
    """libvoge_hip_ab.so: the -DVOGE_AB build (round 3's scalar-sigma sweep + voge_debug_sweep_variant).  A test artefact: the
    product library has neither (no process-wide switch in the stable ABI)."""
    import ctypes
    from voge_amd import _lib
    ctx = _lib.using(_lib.AB_LIB_PATH)
    lib = ctx.__enter__()
    lib.voge_debug_sweep_variant.restype, lib.voge_debug_sweep_variant.argtypes = ctypes.c_int, [ctypes.c_int]
    return ctx, lib


@pytest.mark.parametrize("N,H,W,K,B,r_lo,r_hi,extent", [
    (3000, 72, 88, 40, 1, 0.03, 0.08, 1.0),      # K = 40, ragged image
    (1500, 40, 56, 25, 2, 0.05, 0.12, 1.0),      # odd K (the slot-by-slot epilogue), two views
    (70000, 96, 96, 12, 1, 0.01, 0.03, 1.0),     # more than 65536 Gaussians: list entries carry stream positions
    (20000, 64, 64, 7, 1, 0.004, 0.009, 0.2),    # a small dense object: pooled lists, segment extensions
    (70000, 40, 48, 6, 1, 0.6, 0.7, 0.5),        # > 65536 Gaussians that nearly all hit: the stream-everything fallback's wide form
])
def test_rebuilt_sweep_equals_round_3_sweep_bit_for_bit(hip_lib, N, H, W, K, B, r_lo, r_hi, extent):
    """sweep_iso_kernel (fp32 len + 16-bit handle per list entry, float-compare commits) against trace_fwd_kernel<1, true>
    (64-bit (ord(len), id) keys): the same "K lexicographically smallest (len, id)" (ray_trace_voge.cu:197-212), hence the
    same index lists, hit counts and -- the evaluation being the same operations -- the same len / act / dsd bits, with and
    without act / dsd (round 3's kernel runs in the -DVOGE_AB build of the library, libvoge_hip_ab.so; the last case is the wide,
    two-pass form of a stream of more than 65536 entries -- ADVICE r4)."""
    from voge_amd import ops
    rng = np.random.default_rng(N + K)
    verts = rng.uniform(-extent, extent, (N, 3)).astype(np.float32)
    r = rng.uniform(r_lo, r_hi, N)
    sig = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.0, 3.3][:B], [10.0, -25.0][:B], [20.0, 160.0][:B])
    rays, origin = camera_np.pixel_rays(R, T, 1.1 * W, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    a = np.ascontiguousarray(np.broadcast_to((2 * sig)[None], (B, N))).astype(np.float32)
    thr_act = oracle.thr_act_of(0.01)
    out = {}

    def both():
        full = [n(x) for x in ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)]
        li, ll, lz = ops.trace_lean(1, t(mus.reshape(-1, 3)), t(a.reshape(-1)), None, t(rays), None, thr_act, K)      # (no act / dsd)
        return full, [n(li), n(ll), n(lz.cnt)]
    out[0] = both()                                   # the product library
    ctx, ab = _ab_library()
    try:
        assert ab.voge_debug_sweep_variant(1) == 0    # round 3's sweep, in the A/B build
        out[1] = both()
    finally:
        ab.voge_debug_sweep_variant(0)
        ctx.__exit__()
    assert (out[0][0][0] >= 0).sum() > 500
    for x, y in zip(out[0][0], out[1][0]):
        assert np.array_equal(x, y)
    if out[0][1] is not None:
        for x, y in zip(out[0][1], out[1][1]):
            assert np.array_equal(x, y)
        assert np.array_equal(out[0][1][0], out[0][0][0]) and np.array_equal(out[0][1][1], out[0][0][1])


@pytest.mark.parametrize("N,H,W,K,B,iso_every,diag_every", [
    (3000, 72, 88, 40, 1, 0, 0),       # every Gaussian a full 3x3 form, K = 40, ragged image
    (1500, 40, 56, 25, 2, 3, 0),       # a third of them isotropic (mixed trips), odd K, two views
    (70000, 64, 64, 12, 1, 0, 0),      # more than 65536 Gaussians: list entries carry stream positions
    (3000, 72, 88, 24, 1, 0, 1),       # round 6: every Gaussian a DIAGONAL form (per-axis sigmas): the sweep's diagonal trips
    (2500, 56, 64, 17, 1, 4, 1),       # ... a quarter of them isotropic among diagonal ones (mixed trips inside all-diagonal chunks)
    (2500, 56, 64, 16, 2, 5, 2),       # ... diagonal, full and isotropic forms side by side (chunks that are NOT all-diagonal)
])
def test_general_sweep_equals_round_3_general_sweep_bit_for_bit(hip_lib, N, H, W, K, B, iso_every, diag_every):
    """Round 5: sweep_iso_kernel<true> -- the scalar kernel's design (fp32 len + 16-bit handle per list entry, float-compare
    commits, SoA-staged records, packed evaluation, exit test every 16) for full 3x3 forms -- against round 3's general sweep,
    trace_fwd_kernel<1, false> (64-bit keys), which lives on in the -DVOGE_AB build: the same index lists, hit counts and the
    same len / act / dsd BITS (pair_eval_gen's operations in both), with and without act / dsd."""
    from voge_amd import ops
    rng = np.random.default_rng(N + K + 1)
    verts = rng.uniform(-1.0, 1.0, (N, 3)).astype(np.float32)
    r = rng.uniform(0.03, 0.08, N) * (1.0 if N < 50000 else 0.3)
    L = rng.normal(size=(N, 3, 3)) * 0.35 + np.eye(3)[None]
    S = np.einsum("nij,nkj->nik", L, L) / (r * r / (2 * np.log(1 / 0.6)))[:, None, None]      # SPD, anisotropic
    if diag_every:      # (per-axis forms: the off-diagonal coefficients exactly zero)
        ax = rng.uniform(0.5, 2.0, (N, 3)) / (r * r / (2 * np.log(1 / 0.6)))[:, None]
        D = np.zeros((N, 3, 3))
        D[:, [0, 1, 2], [0, 1, 2]] = ax
        S[::diag_every] = D[::diag_every]
    if iso_every:
        S[::iso_every] = np.eye(3)[None] * (1.0 / (r[::iso_every] ** 2 / (2 * np.log(1 / 0.6))))[:, None, None]
    S = S.astype(np.float32)
    R, T = camera_np.look_at_view_transform([3.0, 3.3][:B], [10.0, -25.0][:B], [20.0, 160.0][:B])
    rays, origin = camera_np.pixel_rays(R, T, 1.1 * W, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = np.ascontiguousarray(np.broadcast_to((2 * S)[None], (B, N, 3, 3))).astype(np.float32)
    thr_act = oracle.thr_act_of(0.01)

    def both():
        full = [n(x) for x in ops.ray_trace_fine(t(mus.reshape(-1, 3)), t(isg.reshape(-1, 3, 3)), t(rays), None, thr_act, 16, K)]
        li, ll, lz = ops.trace_lean(0, t(mus.reshape(-1, 3)), t(isg.reshape(-1, 3, 3)), None, t(rays), None, thr_act, K)      # (no act / dsd)
        return full, [n(li), n(ll), n(lz.cnt)]
    new = both()
    ctx, ab = _ab_library()
    try:
        assert ab.voge_debug_sweep_variant(1) == 0
        old = both()
    finally:
        ab.voge_debug_sweep_variant(0)
        ctx.__exit__()
    assert (new[0][0] >= 0).sum() > 500
    for x, y, name in zip(new[0], old[0], ("idx", "len", "act", "dsd")):
        assert np.array_equal(x, y), name
    for x, y, name in zip(new[1], old[1], ("idx", "len", "cnt")):
        assert np.array_equal(x, y), name
    assert np.array_equal(new[1][0], new[0][0]) and np.array_equal(new[1][1], new[0][1])


@pytest.mark.parametrize("N,extent,r_lo,r_hi,K,what", [
    (70000, 0.5, 0.6, 0.7, 6, "stream-everything fallback with more than 65536 Gaussians: 32-bit handles, two half-tile passes"),
    (90000, 0.05, 0.002, 0.004, 9, "pooled tile lists longer than 65536 entries"),
])
def test_streams_longer_than_16_bit_handles(hip_lib, N, extent, r_lo, r_hi, K, what):
    """The rebuilt sweep keeps a 16-bit handle per list entry; a stream of more than 65536 entries in a scene of more than
    65536 Gaussians takes its wide form (32-bit handles over half the rays at a time).  Against the brute-force oracle."""
    from voge_amd import ops
    rng = np.random.default_rng(N)
    verts = rng.uniform(-extent, extent, (N, 3)).astype(np.float32)
    r = rng.uniform(r_lo, r_hi, N)
    sig = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
    H, W = 40, 48
    R, T = camera_np.look_at_view_transform(3.0, 10.0, 20.0)
    rays, origin = camera_np.pixel_rays(R, T, 110.0, (W / 2.0, H / 2.0), (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(0.01)
    ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
    assert (ref[0][..., 0] >= 0).sum() > 10, what
    a = np.ascontiguousarray(isg[..., 0, 0])
    got = [n(x) for x in ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(a.reshape(-1)), t(rays), None, thr_act, K)]
    used, cap = ops.trace_pool_usage("cuda:0", 1, N, H, W)
    log_line(f"[sweep] {what}: pool used {used} of {cap}")
    compare_trace(got, ref, thr_act, min_match=0.995, label=f"wide handles N={N}")
