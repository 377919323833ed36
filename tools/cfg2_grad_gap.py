"""Where cfg2's (bunny: A ~ 2e4..1e6 at distance 6) vertex-gradient error of 1.9e-4 of scale comes from: the oracle is
substituted stage by stage (run on the GPU box: python tools/cfg2_grad_gap.py).  Every row compares a quantity computed by
ONE HIP stage from the oracle's fp64 inputs (rounded to fp32) with the oracle's own fp64 result for the same inputs; the last
rows give the fp32 reference-order oracle's error for the same chain (the reference CUDA path's own arithmetic floor)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import camera_np
from util import bunny_scene
from voge_amd import ops

DEV = "cuda:0"
t = lambda a, rg=False: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV, requires_grad=rg)
n = lambda x: x.detach().cpu().numpy().astype(np.float64)
sc = bunny_scene()
H, W = sc["image_size"]; K = sc["K"]
R, T = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
rays, origin = camera_np.pixel_rays(R, T, sc["focal"], sc["principal"], (H, W))
verts = np.asarray(sc["verts"], np.float32); sig = np.asarray(sc["sigmas"], np.float32); cols = np.asarray(sc["colors"], np.float64)
mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
a = np.ascontiguousarray(isg[..., 0, 0])
thr_act = oracle.thr_act_of(0.01)
idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, K, thr_act)
w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
rgb = oracle.merge_fwd(cols, idx, w, vn)
img, sil = oracle.blend_fwd(rgb, w)
g_img = np.random.default_rng(2).normal(size=img.shape)
g_rgb = g_img * (rgb + (1 - sil)[..., None] < 1)
g_attr, g_w = oracle.merge_bwd(cols, idx, w, vn, g_rgb)
g_w = g_w - ((g_rgb.sum(-1)) * (w.sum(-1) < 1))[..., None] * (np.arange(K)[None, None, None] < vn[..., None])
g_act, g_len, g_dsd = oracle.composite_bwd(act, ln, dsd, g_w, 1.0)
_, g_mu, g_A = oracle.trace_bwd(mus, isg, rays, idx, g_len, g_act, g_dsd)
g_mu = g_mu.reshape(-1, 3)
scale = max(1.0, np.abs(g_mu).max())
print(f"scene: {len(verts)} Gaussians, |g_mu|max = {scale:.3e}, |g_len|max {np.abs(g_len).max():.3e} |g_act|max {np.abs(g_act).max():.3e} |g_dsd|max {np.abs(g_dsd).max():.3e}")
rel = lambda got, want, s=None: np.abs(got - want).max() / (s or max(1.0, np.abs(want).max()))

# ---- stage B alone: the HIP trace backward on the oracle's fp64 (g_len, g_act, g_dsd), same index lists
tm, ta = t(mus.reshape(-1, 3), True), t(a.reshape(-1), True)
i2, l2, a2, d2 = ops._RayTraceVoGEIso.apply(tm, ta, t(rays), None, thr_act, K)
same = (i2.cpu().numpy() == idx).all(-1)
print(f"index lists identical on {same.mean() * 100:.3f} % of the pixels")
m = same[..., None]
(l2 * t(g_len * m) + a2 * t(g_act * m) + d2 * t(g_dsd * m)).sum().backward()
_, g_mu_m, _ = oracle.trace_bwd(mus, isg, rays, np.where(m, idx, -1), g_len * m, g_act * m, g_dsd * m)
print(f"trace bwd alone (HIP, fp64 upstream grads): g_mu err / scale = {rel(n(tm.grad), g_mu_m.reshape(-1, 3), scale):.2e}")
gf = [x.astype(np.float32).astype(np.float64) for x in (g_len * m, g_act * m, g_dsd * m)]
_, g_mu_r, _ = oracle.trace_bwd(mus, isg, rays, np.where(m, idx, -1), *gf)
print(f"   of which the fp32 rounding of the upstream grads alone: {rel(g_mu_r.reshape(-1, 3), g_mu_m.reshape(-1, 3), scale):.2e}")
_, g_mu_f32, _ = oracle.trace_bwd(mus, isg, rays, np.where(m, idx, -1), g_len * m, g_act * m, g_dsd * m, precision="f32")
print(f"   the fp32 reference-order oracle's trace bwd on the same: {rel(g_mu_f32.reshape(-1, 3).astype(np.float64), g_mu_m.reshape(-1, 3), scale):.2e}")

# ---- stage A alone: the HIP composite backward on the oracle's fp64 inputs
ta_, tl_, td_ = t(act, True), t(ln, True), t(dsd, True)
wq, _ = ops.composite(t(idx).to(torch.int32), ta_, tl_, td_, 1.0)
(wq * t(g_w)).sum().backward()
for name, got, want in (("g_act", ta_.grad, g_act), ("g_len", tl_.grad, g_len), ("g_dsd", td_.grad, g_dsd)):
    live = idx >= 0
    print(f"composite bwd alone: {name} err / max = {rel(n(got)[live], want[live]):.2e}")
# ... pushed through the fp64 trace backward: what the composite's error does to g_mu
_, g_mu_c, _ = oracle.trace_bwd(mus, isg, rays, idx, n(tl_.grad) * (idx >= 0), n(ta_.grad) * (idx >= 0), n(td_.grad) * (idx >= 0))
print(f"composite bwd (HIP) -> trace bwd (fp64): g_mu err / scale = {rel(g_mu_c.reshape(-1, 3), g_mu, scale):.2e}")
# ---- the composite's INPUTS in fp32 (len / act / dsd as the HIP forward produces them) through the fp64 backward chain
ga32, gl32, gd32 = oracle.composite_bwd(n(a2) * m + act * (1 - m), n(l2) * m + ln * (1 - m), n(d2) * m + dsd * (1 - m), g_w, 1.0)
_, g_mu_i, _ = oracle.trace_bwd(mus, isg, rays, idx, gl32, ga32, gd32)
print(f"fp32 forward values (HIP len/act/dsd) -> fp64 composite bwd -> fp64 trace bwd: g_mu err / scale = {rel(g_mu_i.reshape(-1, 3), g_mu, scale):.2e}")
for nm, got, want in (("len", n(l2), ln), ("act", n(a2), act), ("dsd", n(d2), dsd)):
    live = (idx >= 0) & m
    print(f"   HIP forward {nm}: max abs err {np.abs(got - want)[live].max():.3e} (max |value| {np.abs(want[live]).max():.3e})")
# ---- the fp32 reference-order oracle for the whole chain
i32, l32, a32, d32 = oracle.trace_fwd(mus, isg, rays, K, thr_act, precision="f32")
s32 = (i32 == idx).all(-1)[..., None]
w32, vn32 = oracle.composite_fwd(idx, np.where(s32, a32, act), np.where(s32, l32, ln), np.where(s32, d32, dsd), 1.0, precision="f32")
ga, gl, gd = oracle.composite_bwd(np.where(s32, a32, act), np.where(s32, l32, ln), np.where(s32, d32, dsd), g_w, 1.0, precision="f32")
_, g_mu_ref32, _ = oracle.trace_bwd(mus, isg, rays, idx, gl, ga, gd, precision="f32")
print(f"fp32 REFERENCE-ORDER oracle, whole chain (its own len/act/dsd where its lists agree: {s32.mean() * 100:.2f} %): g_mu err / scale = "
      f"{rel(np.asarray(g_mu_ref32, np.float64).reshape(-1, 3), g_mu, scale):.2e};  its act: max abs err {np.abs(a32 - act)[(idx >= 0) & s32].max():.3e}")
# ---- round 5: the FORMAT's floor -- the fp64 chain with len (alone) rounded to the nearest fp32
def chain64(ln_, act_, dsd_):
    w_, vn_ = oracle.composite_fwd(idx, act_, ln_, dsd_, 1.0)
    rgb_ = oracle.merge_fwd(cols, idx, w_, vn_)
    _, sil_ = oracle.blend_fwd(rgb_, w_)
    g_rgb_ = g_img * (rgb_ + (1 - sil_)[..., None] < 1)
    _, gw_ = oracle.merge_bwd(cols, idx, w_, vn_, g_rgb_)
    gw_ = gw_ - ((g_rgb_.sum(-1)) * (w_.sum(-1) < 1))[..., None] * (np.arange(K)[None, None, None] < vn_[..., None])
    ga_, gl_, gd_ = oracle.composite_bwd(act_, ln_, dsd_, gw_, 1.0)
    return oracle.trace_bwd(mus, isg, rays, idx, gl_, ga_, gd_)[1].reshape(-1, 3)
r32 = lambda x: x.astype(np.float32).astype(np.float64)
print(f"fp64 chain, len rounded to NEAREST fp32 (act / dsd fp64): g_mu err / scale = {rel(chain64(r32(ln), act, dsd), g_mu, scale):.3e}   <- the format's floor")
print(f"fp64 chain, act alone rounded: {rel(chain64(ln, r32(act), dsd), g_mu, scale):.2e};  dsd alone rounded: {rel(chain64(ln, act, r32(dsd)), g_mu, scale):.2e}")
