"""The fp32-len FORMAT floor of cfg2's vertex gradient, CPU only: the oracle's fp64 chain with nothing changed but
len (or act, or dsd) rounded to the nearest fp32.  python tools/cfg2_floor.py > profiles/r5_cfg2_floor.txt"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from oracle import camera_np
from util import bunny_scene
sc = bunny_scene()
H, W = sc["image_size"]; K = sc["K"]
R, T = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
rays, origin = camera_np.pixel_rays(R, T, sc["focal"], sc["principal"], (H, W))
verts = np.asarray(sc["verts"], np.float32); sig = np.asarray(sc["sigmas"], np.float32); cols = np.asarray(sc["colors"], np.float64)
mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
thr_act = oracle.thr_act_of(0.01)
t0=time.time()
idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, K, thr_act)
print("trace", time.time()-t0)
w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
rgb = oracle.merge_fwd(cols, idx, w, vn)
img, sil = oracle.blend_fwd(rgb, w)
g_img = np.random.default_rng(2).normal(size=img.shape)
def chain(ln_, act_, dsd_):
    w, vn = oracle.composite_fwd(idx, act_, ln_, dsd_, 1.0)
    rgb = oracle.merge_fwd(cols, idx, w, vn)
    img, sil = oracle.blend_fwd(rgb, w)
    g_rgb = g_img * (rgb + (1 - sil)[..., None] < 1)
    g_attr, g_w = oracle.merge_bwd(cols, idx, w, vn, g_rgb)
    g_w = g_w - ((g_rgb.sum(-1)) * (w.sum(-1) < 1))[..., None] * (np.arange(K)[None, None, None] < vn[..., None])
    g_act, g_len, g_dsd = oracle.composite_bwd(act_, ln_, dsd_, g_w, 1.0)
    _, g_mu, g_A = oracle.trace_bwd(mus, isg, rays, idx, g_len, g_act, g_dsd)
    return g_mu.reshape(-1, 3)
g0 = chain(ln, act, dsd)
scale = max(1.0, np.abs(g0).max())
r32 = lambda x: x.astype(np.float32).astype(np.float64)
def ulp_shift(x, k):      # nearest fp32, then a random error of -k .. +k ulp per value (what a k-ulp-accurate fp32 len looks like)
    f = x.astype(np.float32)
    j = np.random.default_rng(7).integers(-k, k + 1, size=f.shape).astype(np.int32)
    return (f.view(np.int32) + j).view(np.float32).astype(np.float64)
print("scale", scale)
print("len rounded to nearest fp32 only:", np.abs(chain(r32(ln), act, dsd) - g0).max() / scale)
print("len, act, dsd rounded to nearest fp32:", np.abs(chain(r32(ln), r32(act), r32(dsd)) - g0).max() / scale)
print("act only:", np.abs(chain(ln, r32(act), dsd) - g0).max() / scale)
print("dsd only:", np.abs(chain(ln, act, r32(dsd)) - g0).max() / scale)
for k in (1, 2):
    print(f"len nearest fp32 +- {k} ulp (random):", np.abs(chain(ulp_shift(ln, k), act, dsd) - g0).max() / scale)
