"""CPU oracle for the VoGE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  ``voge_amd`` never does and has no CPU fallback.

The arithmetic lives in ``voge_oracle.c`` (plain C, a restatement of
``VoGE/csrc/ray_trace_voge/ray_trace_voge.cu`` and ``VoGE/Aggregation.py``; each function
cites the lines it follows).  This module is the numpy/ctypes face of that library plus
a numpy restatement of the camera/ray conventions (``camera_np``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvoge_oracle.so")
_lib = None


def build(force=False):
    """Compile libvoge_oracle.so with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ("voge_oracle.c", "voge_oracle_impl.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libvoge_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _real(precision):
    if precision == "f64":
        return np.float64
    if precision == "f32":
        return np.float32
    raise ValueError(precision)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def thr_act_of(thr):
    """RayTracing.py:76,85 -- thr_act = -log(thr + 1/inf) with the default-arg inf=1e10."""
    import math
    return -math.log(thr + 1 / 1e10)


def bin_size_of(image_size):
    """RayTracing.py:14-16."""
    return max(int(2 ** np.ceil(np.log2(max(image_size)) - 5)), 10)


def trace_fwd(mus, isg, rays, K, thr_act, bin_points=None, bin_size=0, precision="f64"):
    """mus [B*N,3] or [B,N,3], isg [..,3,3], rays [B,H,W,3] -> idx,len,act,dsd [B,H,W,K]."""
    rays = _c32(rays)
    B, H, W, _ = rays.shape
    mus = _c32(mus).reshape(-1, 3)
    isg = _c32(isg).reshape(-1, 9)
    assert mus.shape[0] == isg.shape[0] and mus.shape[0] % B == 0
    N = mus.shape[0] // B
    T = _real(precision)
    idx = np.empty((B, H, W, K), np.int32)
    ln = np.empty((B, H, W, K), T)
    act = np.empty((B, H, W, K), T)
    dsd = np.empty((B, H, W, K), T)
    BH = BW = M = 0
    if bin_points is not None:
        bin_points = np.ascontiguousarray(bin_points, dtype=np.int32)
        _, BH, BW, M = bin_points.shape
        assert bin_size > 0
    fn = getattr(lib(), "oracle_trace_fwd_" + precision)
    fn(_ptr(mus), _ptr(isg), _ptr(rays), _ptr(bin_points), B, N, H, W, K, BH, BW, M,
       int(bin_size), ctypes.c_double(thr_act), _ptr(idx), _ptr(ln), _ptr(act), _ptr(dsd))
    return idx, ln, act, dsd


def trace_bwd(mus, isg, rays, idx, g_len, g_act, g_dsd, precision="f64"):
    T = _real(precision)
    rays = _c32(rays)
    mus = _c32(mus).reshape(-1, 3)
    isg = _c32(isg).reshape(-1, 9)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    K = idx.shape[-1]
    npix = idx.size // K
    P = mus.shape[0]
    g_len, g_act, g_dsd = (np.ascontiguousarray(g, dtype=T) for g in (g_len, g_act, g_dsd))
    g_ray = np.empty(rays.shape, T)
    g_mus = np.empty((P, 3), T)
    g_isg = np.empty((P, 3, 3), T)
    fn = getattr(lib(), "oracle_trace_bwd_" + precision)
    fn(_ptr(mus), _ptr(isg), _ptr(rays), _ptr(idx), _ptr(g_len), _ptr(g_act), _ptr(g_dsd),
       ctypes.c_long(npix), K, P, _ptr(g_ray), _ptr(g_mus), _ptr(g_isg))
    return g_ray, g_mus, g_isg


def composite_fwd(idx, act, ln, dsd, occ=1.0, precision="f64"):
    T = _real(precision)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    act, ln, dsd = (np.ascontiguousarray(a, dtype=T) for a in (act, ln, dsd))
    K = idx.shape[-1]
    npix = idx.size // K
    w = np.empty(idx.shape, T)
    vn = np.empty(idx.shape[:-1], np.int64)
    fn = getattr(lib(), "oracle_composite_fwd_" + precision)
    fn(_ptr(idx), _ptr(act), _ptr(ln), _ptr(dsd), ctypes.c_double(occ), ctypes.c_long(npix), K,
       _ptr(w), _ptr(vn))
    return w, vn


def composite_bwd(act, ln, dsd, g_weight, occ=1.0, precision="f64"):
    T = _real(precision)
    act, ln, dsd, g_weight = (np.ascontiguousarray(a, dtype=T) for a in (act, ln, dsd, g_weight))
    K = act.shape[-1]
    npix = act.size // K
    g_act, g_len, g_dsd = (np.empty(act.shape, T) for _ in range(3))
    fn = getattr(lib(), "oracle_composite_bwd_" + precision)
    fn(_ptr(act), _ptr(ln), _ptr(dsd), _ptr(g_weight), ctypes.c_double(occ), ctypes.c_long(npix), K,
       _ptr(g_act), _ptr(g_len), _ptr(g_dsd))
    return g_act, g_len, g_dsd


def merge_fwd(attr, idx, weight, valid_num, precision="f64"):
    T = _real(precision)
    attr = np.ascontiguousarray(attr, dtype=T)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    weight = np.ascontiguousarray(weight, dtype=T)
    valid_num = np.ascontiguousarray(valid_num, dtype=np.int64)
    K = idx.shape[-1]
    npix = idx.size // K
    C = attr.shape[1]
    out = np.empty(idx.shape[:-1] + (C,), T)
    fn = getattr(lib(), "oracle_merge_fwd_" + precision)
    fn(_ptr(attr), _ptr(idx), _ptr(weight), _ptr(valid_num), ctypes.c_long(npix), K, C, _ptr(out))
    return out


def merge_bwd(attr, idx, weight, valid_num, g_out, precision="f64"):
    T = _real(precision)
    attr = np.ascontiguousarray(attr, dtype=T)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    weight = np.ascontiguousarray(weight, dtype=T)
    valid_num = np.ascontiguousarray(valid_num, dtype=np.int64)
    g_out = np.ascontiguousarray(g_out, dtype=T)
    K = idx.shape[-1]
    npix = idx.size // K
    Nattr, C = attr.shape
    g_attr = np.empty((Nattr, C), T)
    g_weight = np.empty(idx.shape, T)
    fn = getattr(lib(), "oracle_merge_bwd_" + precision)
    fn(_ptr(attr), _ptr(idx), _ptr(weight), _ptr(valid_num), _ptr(g_out), ctypes.c_long(npix), K, C,
       ctypes.c_long(Nattr), _ptr(g_attr), _ptr(g_weight))
    return g_attr, g_weight


def blend_fwd(rgb, weight, bg=(1.0, 1.0, 1.0), thr=-1.0, precision="f64"):
    T = _real(precision)
    rgb = np.ascontiguousarray(rgb, dtype=T)
    weight = np.ascontiguousarray(weight, dtype=T)
    C = rgb.shape[-1]
    bg = np.ascontiguousarray(np.broadcast_to(np.asarray(bg, dtype=T), (C,)))
    K = weight.shape[-1]
    npix = weight.size // K
    out = np.empty(rgb.shape, T)
    sil = np.empty(weight.shape[:-1], T)
    fn = getattr(lib(), "oracle_blend_fwd_" + precision)
    fn(_ptr(rgb), _ptr(weight), _ptr(bg), ctypes.c_double(thr), ctypes.c_long(npix), K, C,
       _ptr(out), _ptr(sil))
    return out, sil


def render(verts, sigmas, colors, R, T, focal, principal, image_size, K=20, thr=0.01, occ=1.0,
           bg=(1.0, 1.0, 1.0), precision="f64"):
    """Whole forward frame in the oracle: rays (camera_np) -> trace -> composite -> merge -> blend.

    Mirrors GaussianRenderer.forward (Renderer.py:102-150) + to_white_background (:174-176)
    for the max_point_per_bin=-1 candidate set.  verts [N,3], sigmas [N]|[N,3]|[N,3,3],
    R [B,3,3], T [B,3].  Returns a dict of numpy arrays.
    """
    from . import camera_np
    rays, origin = camera_np.pixel_rays(R, T, focal, principal, image_size)
    B = rays.shape[0]
    verts = np.asarray(verts, np.float32)
    mus = (verts[None] - origin[:, None, :].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(np.asarray(sigmas, np.float32))).astype(np.float32)
    isg = np.broadcast_to(isg[None], (B,) + isg.shape)
    idx, ln, act, dsd = trace_fwd(mus, isg, rays, K, thr_act_of(thr), precision=precision)
    w, vn = composite_fwd(idx, act, ln, dsd, occ, precision=precision)
    # indices are b*N+i (RayTracing.py:24-30) -> the attribute table is tiled over the batch
    cols = np.tile(np.asarray(colors), (B, 1))
    rgb = merge_fwd(cols, idx, w, vn, precision=precision)
    img, sil = blend_fwd(rgb, w, bg, precision=precision)
    return dict(rays=rays, origin=origin, mus=mus, isg=isg, idx=idx, len=ln, act=act, dsd=dsd,
                weight=w, valid_num=vn, rgb=rgb, image=img, silhouette=sil)
