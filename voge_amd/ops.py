"""torch.autograd wrappers around the C ABI (include/voge_hip.h).

torch is used here for device memory, streams and autograd bookkeeping only; every
computation on the path is a HIP kernel in libvoge_hip.so.  Ownership follows SURVEY.md §8b:
the caller (these Functions) allocates every output; the library never allocates.
"""
import os

import torch

from . import _lib


# torch._C._cuda_getCurrentRawStream / _cuda_getDevice / _cuda_setDevice are private (present in torch 1.8 ..
# 2.10, the version this image ships); fall back to the public API when a build lacks them.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_set_dev = getattr(torch._C, "_cuda_setDevice", None) or torch.cuda.set_device


def _stream():
    """Raw handle of the current stream of the current device.  (torch.cuda.current_stream() builds a
    Stream object and costs ~10 us a call -- several calls per frame on a path whose small frames are
    host-bound; the raw getter is what torch's own compiled code uses.)"""
    if _raw_stream is not None:
        return _raw_stream(_get_dev())
    return torch.cuda.current_stream().cuda_stream


class _on:
    """`with _on(device)`: make `device` current for the launches inside.  Unlike torch.cuda.device it
    does nothing at all (no guard object, no driver call) when the device is already current."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index if device.index is not None else _get_dev()
        self.prev = -1

    def __enter__(self):
        cur = _get_dev()
        if cur != self.idx:
            self.prev = cur
            _set_dev(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            _set_dev(self.prev)
        return False


def _dev(t, dtype, name):
    """Validate a tensor argument the way the C side cannot: device, dtype, contiguity."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise _lib.VogeHipError(
            f"{name} is on {t.device}: the VoGE hot path runs on a HIP device only (no CPU fallback)")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


_WS = {}


def _workspace(dev, nbytes):
    """Scratch for one entry point: one buffer per (device, stream), grown on demand and reused by later calls.
    Everything the library does is stream-ordered, so consecutive calls on a stream may share their scratch
    (the trace's lists are ~135 MB per 512^2 view: allocating them per call put that much churn on every frame).
    During stream capture the buffer comes from the graph's own pool (a fresh allocation, not cached)."""
    nbytes = max(int(nbytes), 16)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    key = (dev.index if dev.index is not None else _get_dev(), _stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    return ws


def release_workspaces(device=None):
    """Drop the cached scratch buffers (all, or those of one device): the trace's lists are 165 MB per 512^2 view and
    774 MB at 1024^2, held per (device, stream) from the first call on -- a long-lived process that is done rendering big
    frames, or that warmed a side stream for graph capture, gives the memory back here.  The next call re-allocates."""
    idx = None if device is None else (torch.device(device).index if torch.device(device).index is not None else _get_dev())
    for key in [k for k in _WS if idx is None or k[0] == idx]:
        del _WS[key]


class _GeneralPreamble(torch.autograd.Function):
    """Renderer.py:130-137 for (N,3) / (N,3,3) sigmas as one launch each way (voge_general_preamble_fwd / _bwd):
    forward(verts [N,3] | [B,N,3], sigmas [N,3] | [N,3,3] | [B,N,...], origin [B,3]) -> mus [B*N,3], isigmas [B*N,3,3]
    (= verts - origin[b], 2 * expend_sigma(sigmas)).  No gradient for origin (camera optimisation takes the torch chain)."""

    @staticmethod
    def forward(ctx, verts, sigmas, origin):
        lib = _lib.load()
        v_c, s_c, o_c = _dev(verts, torch.float32, "verts"), _dev(sigmas, torch.float32, "sigmas"), _dev(origin, torch.float32, "origin")
        B = o_c.shape[0]
        shared_v = v_c.dim() == 2
        N = v_c.shape[-2]
        kind = 2 if tuple(s_c.shape[-2:]) == (3, 3) and s_c.dim() >= 3 else 1
        shared_s = s_c.dim() == (3 if kind == 2 else 2)
        assert v_c.shape[-1] == 3 and (shared_v or v_c.shape[0] == B) and o_c.shape == (B, 3)
        assert s_c.shape[-1] == 3 and s_c.shape[-(kind + 1)] == N and (shared_s or s_c.shape[0] == B)
        mus = torch.empty((B * N, 3), dtype=torch.float32, device=v_c.device)
        isg = torch.empty((B * N, 3, 3), dtype=torch.float32, device=v_c.device)
        with _on(v_c.device):
            rc = lib.voge_general_preamble_fwd(_p(v_c), _p(s_c), _p(o_c), B, N, int(shared_v), int(shared_s), kind, _p(mus), _p(isg),
                                               _stream())
        _lib.check(rc, "voge_general_preamble_fwd")
        ctx.meta = (B, N, shared_v, shared_s, kind, tuple(v_c.shape), tuple(s_c.shape))
        ctx.set_materialize_grads(False)
        return mus, isg

    @staticmethod
    def backward(ctx, g_mus, g_isg):
        lib = _lib.load()
        B, N, shared_v, shared_s, kind, vshape, sshape = ctx.meta
        if ctx.needs_input_grad[2]:
            raise _lib.VogeHipError("the fused preamble carries no gradient for the camera centre")
        g_v = g_s = None
        gm = None if g_mus is None else _dev(g_mus, torch.float32, "grad_mus")
        ga = None if g_isg is None else _dev(g_isg, torch.float32, "grad_isigmas")
        dev = (gm if gm is not None else ga)
        if dev is None:
            return None, None, None
        dev = dev.device
        if ctx.needs_input_grad[0] and gm is not None:
            g_v = torch.empty(vshape, dtype=torch.float32, device=dev)
        if ctx.needs_input_grad[1] and ga is not None:
            g_s = torch.empty(sshape, dtype=torch.float32, device=dev)
        if g_v is not None or g_s is not None:
            with _on(dev):
                rc = lib.voge_general_preamble_bwd(_p(gm if g_v is not None else None), _p(ga if g_s is not None else None), B, N,
                                                   int(shared_v), int(shared_s), kind, _p(g_v), _p(g_s), _stream())
            _lib.check(rc, "voge_general_preamble_bwd")
        return g_v, g_s, None


def general_preamble(verts, sigmas, origin):
    """-> (mus [B*N,3], isigmas [B*N,3,3]) of the general path, one launch (see _GeneralPreamble)."""
    return _GeneralPreamble.apply(verts, sigmas, origin)


def trace_pool_usage(dev, B, N, H, W):
    """(used, capacity) of the candidate-list pool after the last forward trace of this shape on the current stream of
    `dev` (include/voge_hip.h: voge_trace_pool_usage) -- diagnostics: used > 0 means some 16x16-pixel quads held more
    candidates than the in-LDS sort takes (a small, dense object); used > capacity means the pool ran out and those
    quads fell back to streaming every Gaussian.  Synchronises."""
    import ctypes
    lib = _lib.load()
    dev = torch.device(dev)
    with _on(dev):
        nbytes = lib.voge_trace_workspace_bytes(B, N, H, W)
        ws = _workspace(dev, nbytes)
        used, cap = ctypes.c_int(0), ctypes.c_int(0)
        torch.cuda.current_stream(dev).synchronize()
        _lib.check(lib.voge_trace_pool_usage(_p(ws), nbytes, B, N, H, W, ctypes.byref(used), ctypes.byref(cap)), "voge_trace_pool_usage")
    return used.value, cap.value


def _tag_index(sel_idx, cnt, n_index):
    """Bookkeeping the trace leaves on the index tensor it returns:
    voge_hit_count  (cnt [B,H,W] int32, torch version of sel_idx when it was written): lets aggregation() skip its
                    pass over idx and the backwards skip empty slots.  A later in-place edit of sel_idx through
                    torch (e.g. masking Gaussians out with -1) bumps the version and invalidates the count.
    voge_index_bound  = (number of Gaussians the indices address (B*N), version): merge_final's range assert
                    (Aggregation.py:120) becomes a host-side comparison of two shapes while the tensor is unmodified."""
    if cnt is not None:
        sel_idx.voge_hit_count = (cnt, sel_idx._version)
    sel_idx.voge_index_bound = (int(n_index), sel_idx._version)


def carry_tags(src, dst):
    """A reshaped view of a fragment tensor (Fragments.unsqueeze / squeeze / [0] of a one-view batch / copy) still is
    the same memory: hand the trace's bookkeeping on, so the view keeps the fast paths (no synchronising range check,
    hit counts, the fused backward).  A true slice or a copy gets nothing."""
    if dst is src or not (isinstance(dst, torch.Tensor) and dst.is_cuda and dst.is_contiguous()
                          and dst.data_ptr() == src.data_ptr() and dst.numel() == src.numel() and dst.dtype == src.dtype):
        return dst
    tag = getattr(src, "voge_hit_count", None)
    if tag is not None:
        dst.voge_hit_count = (tag[0].view(dst.shape[:-1]), tag[1])
    for name in ("voge_index_bound", "voge_through"):
        tag = getattr(src, name, None)
        if tag is not None:
            setattr(dst, name, tag)
    return dst


def cones_of(rays, B, H, W):
    """The super-tile cones pixel_rays() left on the ray tensor it returned, if they still describe it (same
    tensor, not modified through torch since), else None -- the trace then derives them itself."""
    tag = getattr(rays, "voge_cones", None)
    if tag is None:
        return None
    cones, version = tag
    if version != rays._version or cones.device != rays.device or tuple(rays.shape) != (B, H, W, 3):
        return None
    return cones


def hit_count_of(sel_idx):
    """The trace's per-pixel hit count if it still describes `sel_idx`, else None."""
    tag = getattr(sel_idx, "voge_hit_count", None)
    if tag is None:
        return None
    cnt, version = tag
    if version != sel_idx._version or cnt.shape != sel_idx.shape[:-1] or cnt.device != sel_idx.device:
        return None
    return cnt


def check_index_range(idx, n_attr):
    """merge_final's `assert vert_attr.shape[0] > vert_assign.max()` (Aggregation.py:120).  Indices written by
    the trace carry their range (voge_index_bound = B*N) while nobody has edited them through torch: a host comparison
    (stricter than the reference's: it fails whenever the table is shorter than B*N rows, hit or not -- INTEGRATION.md
    section 4).  Any other index tensor is checked on the device as the reference does (a synchronising reduction),
    except during stream capture."""
    tag = getattr(idx, "voge_index_bound", None)
    if tag is not None and tag[1] == idx._version:
        if n_attr < tag[0]:
            raise AssertionError(
                f"vert_attr has {n_attr} rows but the fragments index {tag[0]} Gaussians (a batch of B views addresses "
                f"rows b*N+n: tile the attributes over the batch, as the reference requires -- Aggregation.py:120)")
        return
    if idx.numel() and not torch.cuda.is_current_stream_capturing():
        assert n_attr > int(idx.max()), "vert_attr.shape[0] must exceed vert_assign.max() (Aggregation.py:120)"


class _RayTraceVoGE(torch.autograd.Function):
    """Mirror of VoGE/RayTracing.py:154-206 (`_RayTraceVoGE`).

    forward(mus [P,3], isigmas [P,3,3], rays [B,H,W,3], bin_points, thr_act, bin_size, n_assign)
    -> sel_idx int32, sel_len, sel_act, sel_dsd, each [B,H,W,K].
    bin_points: None -> every Gaussian of the batch element is a candidate (the
    max_points_per_bin == -1 list of RayTracing.py:22-26, never materialised); a tensor
    [B,BH,BW,M] -> explicit candidate lists as in the reference; a tensor [B,3] of dtype
    float -> "all candidates in front of the camera" (view axis per batch element).
    """

    @staticmethod
    def forward(ctx, mus, isigmas, rays, bin_points, thr_act, bin_size, n_assign):
        lib = _lib.load()
        mus_c = _dev(mus, torch.float32, "mus")
        isg_c = _dev(isigmas, torch.float32, "isigmas")
        rays_c = _dev(rays, torch.float32, "rays")
        assert mus_c.dim() == 2 and mus_c.shape[1] == 3
        assert isg_c.dim() == 3 and isg_c.shape[1:] == (3, 3) and isg_c.shape[0] == mus_c.shape[0]
        assert rays_c.dim() == 4 and rays_c.shape[3] == 3
        B, H, W, _ = rays_c.shape
        P = mus_c.shape[0]
        K = int(n_assign)
        dev = rays_c.device
        sel_idx = torch.empty((B, H, W, K), dtype=torch.int32, device=dev)
        sel_len = torch.empty((B, H, W, K), dtype=torch.float32, device=dev)
        sel_act = torch.empty_like(sel_len)
        sel_dsd = torch.empty_like(sel_len)
        # per-pixel hit count: tells the backward (and aggregation) which slots are filled, so the index tensor
        # may later be rewritten in place by merge_final (-1 -> 0, Aggregation.py:131) without a defensive copy
        cnt = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        with _on(dev):
            if bin_points is not None and bin_points.dtype in (torch.int32, torch.int64):
                bins = _dev(bin_points, torch.int32, "bin_points")
                assert bins.dim() == 4 and bins.shape[0] == B
                rc = lib.voge_trace_topk_list_fwd(
                    _p(mus_c), _p(isg_c), _p(rays_c), _p(bins), B, P, H, W, K, bins.shape[1], bins.shape[2],
                    bins.shape[3], int(bin_size), float(thr_act), _p(sel_idx), _p(sel_len), _p(sel_act),
                    _p(sel_dsd), _p(cnt), _stream())
                _lib.check(rc, "voge_trace_topk_list_fwd")
            else:
                assert B > 0 and P % B == 0, "mus must hold B*N rows"
                N = P // B
                fwd = None if bin_points is None else _dev(bin_points, torch.float32, "cam_fwd")
                nbytes = lib.voge_trace_workspace_bytes(B, N, H, W)
                ws = _workspace(dev, nbytes)
                rc = lib.voge_trace_topk_fwd(
                    _p(mus_c), _p(isg_c), _p(rays_c), _p(fwd), _p(cones_of(rays_c, B, H, W)), B, N, H, W, K, float(thr_act), _p(ws), nbytes,
                    _p(sel_idx), _p(sel_len), _p(sel_act), _p(sel_dsd), _p(cnt), _stream())
                _lib.check(rc, "voge_trace_topk_fwd")
        ctx.save_for_backward(mus_c, isg_c, rays_c)
        ctx.sel_idx = sel_idx
        ctx.cnt = cnt
        _tag_index(sel_idx, cnt, P)
        ctx.mark_non_differentiable(sel_idx)
        ctx.set_materialize_grads(False)
        return sel_idx, sel_len, sel_act, sel_dsd

    @staticmethod
    def backward(ctx, grad_sel_idx, grad_sel_len, grad_sel_act, grad_sel_dsd):
        lib = _lib.load()
        mus, isg, rays = ctx.saved_tensors
        sel_idx = ctx.sel_idx
        B, H, W, K = sel_idx.shape
        P = mus.shape[0]
        zeros = None

        def g(t):
            nonlocal zeros
            if t is None:
                if zeros is None:
                    zeros = torch.zeros(sel_idx.shape, dtype=torch.float32, device=sel_idx.device)
                return zeros
            return _dev(t, torch.float32, "grad")
        gl, ga, gd = g(grad_sel_len), g(grad_sel_act), g(grad_sel_dsd)
        g_ray = torch.empty_like(rays) if ctx.needs_input_grad[2] else None
        g_mus = torch.empty_like(mus)
        g_isg = torch.empty_like(isg)
        with _on(rays.device):
            nbytes = lib.voge_trace_bwd_workspace_bytes(P)
            ws = _workspace(rays.device, nbytes)
            rc = lib.voge_trace_bwd(_p(mus), _p(isg), _p(rays), _p(sel_idx), _p(ctx.cnt), _p(gl), _p(ga), _p(gd), P,
                                    B * H, W, K, _p(ws), nbytes, _p(g_ray), _p(g_mus), _p(g_isg), _stream())
        _lib.check(rc, "voge_trace_bwd")
        return g_mus, g_isg, g_ray, None, None, None, None


class _RayTraceVoGEIso(torch.autograd.Function):
    """The same trace for isotropic Gaussians given as ONE scalar each (A = a I): what
    expend_sigma((N,)) followed by 2*sigma (Aggregation.py:155-157, Renderer.py:133) describes.
    forward(mus [P,3], a [P], rays [B,H,W,3], cam_fwd | None, thr_act, n_assign); the backward
    returns the gradient of the scalar directly (four sums per Gaussian instead of twelve)."""

    @staticmethod
    def forward(ctx, mus, a, rays, cam_fwd, thr_act, n_assign):
        lib = _lib.load()
        mus_c = _dev(mus, torch.float32, "mus")
        a_c = _dev(a, torch.float32, "a")
        rays_c = _dev(rays, torch.float32, "rays")
        assert mus_c.dim() == 2 and mus_c.shape[1] == 3 and a_c.dim() == 1 and a_c.shape[0] == mus_c.shape[0]
        assert rays_c.dim() == 4 and rays_c.shape[3] == 3
        B, H, W, _ = rays_c.shape
        P = mus_c.shape[0]
        assert B > 0 and P % B == 0, "mus must hold B*N rows"
        N, K, dev = P // B, int(n_assign), rays_c.device
        sel_idx = torch.empty((B, H, W, K), dtype=torch.int32, device=dev)
        sel_len = torch.empty((B, H, W, K), dtype=torch.float32, device=dev)
        sel_act = torch.empty_like(sel_len)
        sel_dsd = torch.empty_like(sel_len)
        cnt = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        fwd = None if cam_fwd is None else _dev(cam_fwd, torch.float32, "cam_fwd")
        with _on(dev):
            nbytes = lib.voge_trace_workspace_bytes(B, N, H, W)
            ws = _workspace(dev, nbytes)
            rc = lib.voge_trace_topk_fwd_iso(
                _p(mus_c), _p(a_c), _p(rays_c), _p(fwd), _p(cones_of(rays_c, B, H, W)), B, N, H, W, K, float(thr_act), _p(ws), nbytes,
                _p(sel_idx), _p(sel_len), _p(sel_act), _p(sel_dsd), _p(cnt), _stream())
        _lib.check(rc, "voge_trace_topk_fwd_iso")
        ctx.save_for_backward(mus_c, a_c, rays_c)
        ctx.sel_idx = sel_idx      # may later be rewritten in place by merge_final; cnt marks the filled slots
        ctx.cnt = cnt
        _tag_index(sel_idx, cnt, P)
        ctx.mark_non_differentiable(sel_idx)
        ctx.set_materialize_grads(False)
        return sel_idx, sel_len, sel_act, sel_dsd

    @staticmethod
    def backward(ctx, grad_sel_idx, grad_sel_len, grad_sel_act, grad_sel_dsd):
        lib = _lib.load()
        mus, a, rays = ctx.saved_tensors
        sel_idx = ctx.sel_idx
        B, H, W, K = sel_idx.shape
        P = mus.shape[0]
        zeros = None

        def g(t):
            nonlocal zeros
            if t is None:
                if zeros is None:
                    zeros = torch.zeros(sel_idx.shape, dtype=torch.float32, device=sel_idx.device)
                return zeros
            return _dev(t, torch.float32, "grad")
        gl, ga, gd = g(grad_sel_len), g(grad_sel_act), g(grad_sel_dsd)
        g_ray = torch.empty_like(rays) if ctx.needs_input_grad[2] else None
        g_mus = torch.empty_like(mus)
        g_a = torch.empty_like(a)
        with _on(rays.device):
            nbytes = lib.voge_trace_bwd_iso_workspace_bytes(P)
            ws = _workspace(rays.device, nbytes)
            rc = lib.voge_trace_bwd_iso(_p(mus), _p(a), _p(rays), _p(sel_idx), _p(ctx.cnt), _p(gl), _p(ga), _p(gd), P,
                                        B * H, W, K, _p(ws), nbytes, _p(g_ray), _p(g_mus), _p(g_a), _stream())
        _lib.check(rc, "voge_trace_bwd_iso")
        return g_mus, g_a, g_ray, None, None, None


class _RayTraceVoGEIsoView(torch.autograd.Function):
    """_RayTraceVoGEIso with the renderer's elementwise preamble folded into the kernels
    (Renderer.py:130-137): forward(verts [N,3] | [B,N,3], sigmas [N] | [B,N], origin [B,3], rays [B,H,W,3],
    cam_fwd | None, thr_act, n_assign, sigma_mode) where the Gaussian seen by view b is
    (verts - origin[b], a = 2 sigma (mode 1) | 2 / sigma (mode 2) | sigma (mode 0)).  No gradient for
    origin: callers that optimise the camera use the unfused form."""

    @staticmethod
    def forward(ctx, verts, sigmas, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode):
        lib = _lib.load()
        v_c = _dev(verts, torch.float32, "verts")
        s_c = _dev(sigmas, torch.float32, "sigmas")
        o_c = _dev(origin, torch.float32, "origin")
        rays_c = _dev(rays, torch.float32, "rays")
        assert rays_c.dim() == 4 and rays_c.shape[3] == 3
        B, H, W, _ = rays_c.shape
        shared = v_c.dim() == 2
        assert v_c.shape[-1] == 3 and (shared or v_c.shape[0] == B) and s_c.shape == v_c.shape[:-1]
        assert o_c.shape == (B, 3)
        N, K, dev = v_c.shape[-2], int(n_assign), rays_c.device
        sel_idx = torch.empty((B, H, W, K), dtype=torch.int32, device=dev)
        sel_len = torch.empty((B, H, W, K), dtype=torch.float32, device=dev)
        sel_act = torch.empty_like(sel_len)
        sel_dsd = torch.empty_like(sel_len)
        cnt = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        fwd = None if cam_fwd is None else _dev(cam_fwd, torch.float32, "cam_fwd")
        with _on(dev):
            nbytes = lib.voge_trace_workspace_bytes(B, N, H, W)
            ws = _workspace(dev, nbytes)
            rc = lib.voge_trace_topk_fwd_iso_view(
                _p(v_c), _p(s_c), _p(o_c), int(shared), int(sigma_mode), _p(rays_c), _p(fwd), _p(cones_of(rays_c, B, H, W)), B, N, H, W, K,
                float(thr_act), _p(ws), nbytes, _p(sel_idx), _p(sel_len), _p(sel_act), _p(sel_dsd), _p(cnt), _stream())
        _lib.check(rc, "voge_trace_topk_fwd_iso_view")
        ctx.save_for_backward(v_c, s_c, o_c, rays_c)
        ctx.sel_idx, ctx.cnt, ctx.mode, ctx.shared = sel_idx, cnt, int(sigma_mode), shared
        _tag_index(sel_idx, cnt, B * N)
        ctx.mark_non_differentiable(sel_idx)
        ctx.set_materialize_grads(False)
        return sel_idx, sel_len, sel_act, sel_dsd

    @staticmethod
    def backward(ctx, grad_sel_idx, grad_sel_len, grad_sel_act, grad_sel_dsd):
        lib = _lib.load()
        verts, sigmas, origin, rays = ctx.saved_tensors
        if ctx.needs_input_grad[2]:
            raise _lib.VogeHipError("the fused view form has no gradient for the camera centre; "
                                    "use ray_tracing_iso on centred vertices")
        sel_idx = ctx.sel_idx
        B, H, W, K = sel_idx.shape
        N = verts.shape[-2]
        zeros = None

        def g(t):
            nonlocal zeros
            if t is None:
                if zeros is None:
                    zeros = torch.zeros(sel_idx.shape, dtype=torch.float32, device=sel_idx.device)
                return zeros
            return _dev(t, torch.float32, "grad")
        gl, ga, gd = g(grad_sel_len), g(grad_sel_act), g(grad_sel_dsd)
        g_ray = torch.empty_like(rays) if ctx.needs_input_grad[3] else None
        g_verts = torch.empty_like(verts)
        g_sig = torch.empty_like(sigmas)
        with _on(rays.device):
            nbytes = lib.voge_trace_bwd_iso_workspace_bytes(B * N)
            ws = _workspace(rays.device, nbytes)
            rc = lib.voge_trace_bwd_iso_view(_p(verts), _p(sigmas), _p(origin), int(ctx.shared), ctx.mode, _p(rays),
                                             _p(sel_idx), _p(ctx.cnt), _p(gl), _p(ga), _p(gd), B, N, B * H, W, K,
                                             _p(ws), nbytes, _p(g_ray), _p(g_verts), _p(g_sig), _stream())
        _lib.check(rc, "voge_trace_bwd_iso_view")
        return g_verts, g_sig, None, g_ray, None, None, None, None


# VOGE_THREE_KERNEL_BACKWARD=1: _Fragments.backward runs voge_composite_bwd + voge_trace_bwd* (the round-1 chain, with
# act / dsd materialised and g_len / g_act / g_dsd exchanged through memory) instead of the one-pass voge_fragment_bwd*.
THREE_KERNEL_BACKWARD = os.environ.get("VOGE_THREE_KERNEL_BACKWARD", "0") == "1"


def _grad_weight_layout(g, K):
    """How voge_fragment_bwd* should read a gradient of the weights: (tensor, stride_pix, stride_k) in elements.
    Contiguous [.., K] -> (K, 1); a per-pixel value expanded over the slots (what _Silhouette.backward returns when
    nothing else consumed the weights) -> (1, 0), read in place; anything else is made contiguous."""
    if g is None:
        return None, 0, 0
    if g.dtype == torch.float32 and g.is_cuda:
        if g.is_contiguous():
            return g, K, 1
        if g.stride(-1) == 0 and g[..., 0].is_contiguous():
            return g, 1, 0
    return _dev(g, torch.float32, "grad_weight"), K, 1


class _Fragments(torch.autograd.Function):
    """Trace + composite in one launch chain (voge_fragments_fwd*): what GaussianRenderer.forward needs from
    ray_tracing (RayTracing.py:12-30) followed by aggregation (Aggregation.py:82-107).
    forward(mode, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode, occ)
        -> weight, sel_idx, valid_num, sel_len | cnt, records, act, dsd  (the call's bookkeeping, non-differentiable)
      mode 0: p0 = mus [P,3], p1 = isigmas [P,3,3]           (general 3x3 forms)
      mode 1: p0 = mus [P,3], p1 = a [P]                     (A = a I)
      mode 2: p0 = verts [N,3] | [B,N,3], p1 = sigmas [N] | [B,N], origin [B,3]   (the renderer's preamble folded in)
    The backward is ONE pass over the fragments (voge_fragment_bwd*: the composite's closed form and the trace's chain
    rule, nothing exchanged through memory) for whatever gradient reaches the weights and vert_hit_length -- the
    reference's training loops differentiate through interpolate_attr + get_silhouette (demo/ShapeFitting.py:217,295).
    Only a caller that needs the gradient of the RAYS takes the stand-alone composite + trace backward kernels."""

    @staticmethod
    def forward(ctx, mode, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode, occ):
        lib = _lib.load()
        p0_c, p1_c = _dev(p0, torch.float32, "means"), _dev(p1, torch.float32, "sigmas")
        rays_c = _dev(rays, torch.float32, "rays")
        assert rays_c.dim() == 4 and rays_c.shape[3] == 3
        B, H, W, _ = rays_c.shape
        K, dev = int(n_assign), rays_c.device
        o_c, shared = None, False
        if mode == 2:
            o_c = _dev(origin, torch.float32, "origin")
            shared = p0_c.dim() == 2
            assert p0_c.shape[-1] == 3 and (shared or p0_c.shape[0] == B) and p1_c.shape == p0_c.shape[:-1]
            assert o_c.shape == (B, 3)
            N = p0_c.shape[-2]
        else:
            assert p0_c.dim() == 2 and p0_c.shape[1] == 3 and p0_c.shape[0] % max(B, 1) == 0
            assert p1_c.shape[0] == p0_c.shape[0] and (p1_c.dim() == 1 if mode == 1 else p1_c.shape[1:] == (3, 3))
            N = p0_c.shape[0] // B
        sel_idx = torch.empty((B, H, W, K), dtype=torch.int32, device=dev)
        sel_len = torch.empty((B, H, W, K), dtype=torch.float32, device=dev)
        weight = torch.empty_like(sel_len)
        # scalar sigmas: act / dsd are not materialised at all -- every consumer re-derives them from the records
        # (composite forward, fused backward) or asks for them once (_act_dsd: the three-kernel backward)
        lean = (mode != 0 and B * N < (1 << 26)      # (32-bit byte offsets of the record gathers)
                and os.environ.get("VOGE_FRAGMENTS_KEEP_ACT_DSD", "0") != "1")
        sel_act = None if lean else torch.empty_like(sel_len)
        sel_dsd = None if lean else torch.empty_like(sel_len)
        cnt = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        valid = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        # scalar forms: the per-Gaussian (centred mean, a) records outlive the call -- the fused backward reads them
        records = torch.empty((B * N, 4), dtype=torch.float32, device=dev) if mode != 0 else None
        fwd = None if cam_fwd is None else _dev(cam_fwd, torch.float32, "cam_fwd")
        with _on(dev):
            nbytes = lib.voge_trace_workspace_bytes(B, N, H, W)
            ws = _workspace(dev, nbytes)
            tail = (B, N, H, W, K, float(thr_act), float(occ), _p(ws), nbytes, _p(sel_idx), _p(sel_len), _p(sel_act),
                    _p(sel_dsd), _p(cnt), _p(weight), _p(valid))
            cones = _p(cones_of(rays_c, B, H, W))
            if mode == 2:
                rc = lib.voge_fragments_fwd_iso_view(_p(p0_c), _p(p1_c), _p(o_c), int(shared), int(sigma_mode), _p(rays_c),
                                                     _p(fwd), cones, *tail, _p(records), _stream())
            elif mode == 1:
                rc = lib.voge_fragments_fwd_iso(_p(p0_c), _p(p1_c), _p(rays_c), _p(fwd), cones, *tail, _p(records), _stream())
            else:
                rc = lib.voge_fragments_fwd(_p(p0_c), _p(p1_c), _p(rays_c), _p(fwd), cones, *tail, _stream())
        _lib.check(rc, "voge_fragments_fwd")
        ctx.save_for_backward(p0_c, p1_c, rays_c, sel_len, weight)
        # (only tensors that are NOT differentiable outputs of this node may hang on ctx as plain attributes: an output
        # with a grad_fn kept here would close the cycle ctx -> tensor -> grad_fn -> ctx and leave the frame's buffers to
        # the cyclic garbage collector)
        ctx.origin, ctx.sel_idx, ctx.cnt, ctx.records = o_c, sel_idx, cnt, records
        ctx.act_dsd = [sel_act, sel_dsd]
        ctx.meta = (int(mode), int(sigma_mode), bool(shared), float(occ), B, N)
        _tag_index(sel_idx, cnt, B * N)
        ctx.idx_version = sel_idx._version      # (an index list edited through torch afterwards may hold a Gaussian TWICE in a pixel)
        ctx.mark_non_differentiable(sel_idx, valid, cnt)
        if records is not None:
            ctx.mark_non_differentiable(records)
        if sel_act is not None:
            ctx.mark_non_differentiable(sel_act, sel_dsd)
        ctx.set_materialize_grads(False)
        return weight, sel_idx, valid, sel_len, cnt, records, sel_act, sel_dsd

    @staticmethod
    def backward(ctx, g_weight, _g_idx, _g_valid, g_hitlen, *_unused):
        lib = _lib.load()
        p0, p1, rays, ln, weight = ctx.saved_tensors
        mode, sigma_mode, shared, occ, B, N = ctx.meta
        if g_weight is None and g_hitlen is None:      # nothing reached the fragments (e.g. _ShadeThrough took the frame)
            return (None,) * 10
        sel_idx, cnt = ctx.sel_idx, ctx.cnt
        _, H, W, K = sel_idx.shape
        npix = B * H * W
        dev = rays.device
        if mode == 2 and ctx.needs_input_grad[3]:
            raise _lib.VogeHipError("the fused view form has no gradient for the camera centre; "
                                    "use ray_tracing_iso on centred vertices")
        g0, g1 = torch.empty_like(p0), torch.empty_like(p1)
        # (the one-pass kernel adds a pixel's slots to its table without arbitration -- a pixel's Gaussians are distinct
        #  as the trace wrote them.  A list somebody edited through torch since is no longer known to be: it takes the stand-alone
        #  kernels, whose table elects one writer per key)
        edited = sel_idx._version != ctx.idx_version
        if not (ctx.needs_input_grad[4] or THREE_KERNEL_BACKWARD or B * N >= (1 << 26) or edited):
            # ONE pass: composite backward + trace backward per slot in registers, per-Gaussian sums in a wave-private
            # table (fragment_bwd.hip, SRC = 1).  No act / dsd arrays, no g_len / g_act / g_dsd.
            gw, gs_pix, gs_k = _grad_weight_layout(g_weight, K)
            gh = None if g_hitlen is None else _dev(g_hitlen, torch.float32, "grad_hit_length")
            act, dsd = ctx.act_dsd
            with _on(dev):
                nbytes = lib.voge_fragment_bwd_workspace_bytes(B * N)
                ws = _workspace(dev, nbytes)
                if mode == 0:
                    rc = lib.voge_fragment_bwd(_p(p0), _p(p1), _p(rays), _p(sel_idx), _p(cnt), _p(weight), _p(act), _p(ln), _p(dsd),
                                               _p(gw), gs_pix, gs_k, _p(gh), occ, B * N, B * H, W, K, _p(ws), nbytes, _p(g0), _p(g1),
                                               _stream())
                else:
                    rc = lib.voge_fragment_bwd_iso(_p(ctx.records), _p(p1), int(shared), sigma_mode, _p(rays), _p(sel_idx), _p(cnt),
                                                   _p(weight), _p(act), _p(ln), _p(dsd), _p(gw), gs_pix, gs_k, _p(gh), occ, B, N,
                                                   B * H, W, K, _p(ws), nbytes, _p(g0), _p(g1), _stream())
            _lib.check(rc, "voge_fragment_bwd")
            return None, g0, g1, None, None, None, None, None, None, None
        act, dsd = _act_dsd(ctx.act_dsd, ctx.records, rays, sel_idx, ln, cnt, B * N)
        g_act, g_len, g_dsd = torch.empty_like(act), torch.empty_like(act), torch.empty_like(act)
        with _on(dev):
            if g_weight is None:
                g_act.zero_(); g_len.zero_(); g_dsd.zero_()
            else:
                gw = _dev(g_weight, torch.float32, "grad_weight")
                rc = lib.voge_composite_bwd(_p(act), _p(ln), _p(dsd), _p(weight), _p(cnt), _p(gw), occ, npix, K, _p(g_act),
                                            _p(g_len), _p(g_dsd), _stream())
                _lib.check(rc, "voge_composite_bwd")
            if g_hitlen is not None:      # vert_hit_length is the trace's len itself (Aggregation.py:107)
                g_len = g_len + _dev(g_hitlen, torch.float32, "grad_hit_length")
            g_ray = torch.empty_like(rays) if ctx.needs_input_grad[4] else None
            if mode == 2:
                nbytes = lib.voge_trace_bwd_iso_workspace_bytes(B * N)
                ws = _workspace(dev, nbytes)
                rc = lib.voge_trace_bwd_iso_view(_p(p0), _p(p1), _p(ctx.origin), int(shared), sigma_mode, _p(rays), _p(sel_idx),
                                                 _p(cnt), _p(g_len), _p(g_act), _p(g_dsd), B, N, B * H, W, K, _p(ws), nbytes,
                                                 _p(g_ray), _p(g0), _p(g1), _stream())
            elif mode == 1:
                nbytes = lib.voge_trace_bwd_iso_workspace_bytes(B * N)
                ws = _workspace(dev, nbytes)
                rc = lib.voge_trace_bwd_iso(_p(p0), _p(p1), _p(rays), _p(sel_idx), _p(cnt), _p(g_len), _p(g_act), _p(g_dsd), B * N,
                                            B * H, W, K, _p(ws), nbytes, _p(g_ray), _p(g0), _p(g1), _stream())
            else:
                nbytes = lib.voge_trace_bwd_workspace_bytes(B * N)
                ws = _workspace(dev, nbytes)
                rc = lib.voge_trace_bwd(_p(p0), _p(p1), _p(rays), _p(sel_idx), _p(cnt), _p(g_len), _p(g_act), _p(g_dsd), B * N,
                                        B * H, W, K, _p(ws), nbytes, _p(g_ray), _p(g0), _p(g1), _stream())
        _lib.check(rc, "voge_trace_bwd")
        return None, g0, g1, None, g_ray, None, None, None, None, None


def _act_dsd(cell, records, rays, idx, ln, cnt, P):
    """act / dsd of a fragments() call: what it kept (cell = [act, dsd]), or (scalar sigmas) derived now from its records
    -- once; the arrays then stay in the cell."""
    if cell[0] is None:
        lib = _lib.load()
        act, dsd = torch.empty_like(ln), torch.empty_like(ln)
        K = idx.shape[-1]
        with _on(ln.device):
            rc = lib.voge_fragment_act_dsd_iso(_p(records), _p(rays), _p(idx), _p(ln), _p(cnt), idx.numel() // K, K, P,
                                               _p(act), _p(dsd), _stream())
        _lib.check(rc, "voge_fragment_act_dsd_iso")
        cell[0], cell[1] = act, dsd
    return cell[0], cell[1]


def fragments(mode, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode=0, occ=1.0):
    """-> weight, sel_idx, valid_num, sel_len.  The weight tensor carries `voge_through`: the inputs and bookkeeping
    of this call (detached: nothing in it has a grad_fn), which lets a later to_colored_background(fragments, colors)
    run shade + composite + trace backward as ONE kernel (_ShadeThrough)."""
    weight, sel_idx, valid, sel_len, cnt, records, sel_act, sel_dsd = _Fragments.apply(
        mode, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode, occ)
    if (p0.requires_grad or p1.requires_grad) and torch.is_grad_enabled():
        B = rays.shape[0]
        N = p0.shape[-2] if mode == 2 else p0.shape[0] // max(B, 1)
        rays_d, len_d = rays.detach(), sel_len.detach()      # (same storage and version counter, no grad_fn)
        weight.voge_through = dict(
            mode=int(mode), sigma_mode=int(sigma_mode), shared=bool(mode == 2 and p0.dim() == 2), occ=float(occ), B=B, N=N,
            records=records, rays=rays_d, act=sel_act, dsd=sel_dsd, len=len_d, cnt=cnt, idx=sel_idx,
            sigmas=p1.detach(), means=p0.detach(), p0=p0, p1=p1, rays_requires_grad=bool(rays.requires_grad),
            versions=(weight._version, len_d._version, rays_d._version, sel_idx._version))
    return weight, sel_idx, valid, sel_len


# ----------------------------------------------------------------------------------------------------------------------
# Deferred composite (scalar sigmas).  GaussianRenderer hands its Fragments back before anybody knows what they will be
# used for; the sweep's outputs (index, len, hit count, records) are all the later stages need.  With VOGE_LAZY_COMPOSITE
# (default on) the renderer therefore stops behind the sweep (_TraceLean) and the composite runs when the fragments'
# weights are first asked for:
#   * to_colored_background on untouched fragments -> _CompositeShade: weights AND image in ONE pass (the composite kernel
#     gathers the colours of the slots it holds anyway; nothing is read back);
#   * any other access (frag.vert_weight, interpolate_attr, get_silhouette, ...) -> _CompositeLean: the composite kernel alone.
# Results are the same bits as the eager chain's weights; the image's sums are associated per lane group instead of per
# DPP row (within 1e-7 of the separate kernel).  Both nodes take the Gaussians as inputs and hand them the gradient of
# everything below (composite + trace backward in one pass), like _ShadeThrough.
LAZY_COMPOSITE = os.environ.get("VOGE_LAZY_COMPOSITE", "1") != "0"


class _TraceLean(torch.autograd.Function):
    """The fine trace alone, without act / dsd (voge_fragments_fwd_iso* in trace-only form):
    forward(mode 1 | 2, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode) -> sel_idx, sel_len, cnt, records.
    backward: only vert_hit_length can carry a gradient here (voge_fragment_bwd_iso with g_weight = NULL)."""

    @staticmethod
    def forward(ctx, mode, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode):
        lib = _lib.load()
        p0_c, p1_c = _dev(p0, torch.float32, "means"), _dev(p1, torch.float32, "sigmas")
        rays_c = _dev(rays, torch.float32, "rays")
        assert rays_c.dim() == 4 and rays_c.shape[3] == 3 and mode in (0, 1, 2)
        B, H, W, _ = rays_c.shape
        K, dev = int(n_assign), rays_c.device
        o_c, shared = None, False
        if mode == 0:      # full 3x3 forms: (mus [P,3], isigmas [P,3,3]); records = the packed (mu, A)
            assert p0_c.dim() == 2 and p0_c.shape[1] == 3 and p0_c.shape[0] % max(B, 1) == 0
            assert p1_c.shape == (p0_c.shape[0], 3, 3)
            N = p0_c.shape[0] // B
        elif mode == 2:
            o_c = _dev(origin, torch.float32, "origin")
            shared = p0_c.dim() == 2
            assert p0_c.shape[-1] == 3 and (shared or p0_c.shape[0] == B) and p1_c.shape == p0_c.shape[:-1]
            assert o_c.shape == (B, 3)
            N = p0_c.shape[-2]
        else:
            assert p0_c.dim() == 2 and p0_c.shape[1] == 3 and p0_c.shape[0] % max(B, 1) == 0
            assert p1_c.shape[0] == p0_c.shape[0] and p1_c.dim() == 1
            N = p0_c.shape[0] // B
        sel_idx = torch.empty((B, H, W, K), dtype=torch.int32, device=dev)
        sel_len = torch.empty((B, H, W, K), dtype=torch.float32, device=dev)
        cnt = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        records = torch.empty((B * N, 12 if mode == 0 else 4), dtype=torch.float32, device=dev)
        fwd = None if cam_fwd is None else _dev(cam_fwd, torch.float32, "cam_fwd")
        with _on(dev):
            nbytes = lib.voge_trace_workspace_bytes(B, N, H, W)
            ws = _workspace(dev, nbytes)
            tail = (B, N, H, W, K, float(thr_act), 1.0, _p(ws), nbytes, _p(sel_idx), _p(sel_len), None, None, _p(cnt), None, None)
            cones = _p(cones_of(rays_c, B, H, W))
            if mode == 0:
                rc = lib.voge_trace_lean_fwd(_p(p0_c), _p(p1_c), _p(rays_c), _p(fwd), cones, B, N, H, W, K, float(thr_act), _p(ws),
                                             nbytes, _p(sel_idx), _p(sel_len), _p(cnt), _p(records), _stream())
            elif mode == 2:
                rc = lib.voge_fragments_fwd_iso_view(_p(p0_c), _p(p1_c), _p(o_c), int(shared), int(sigma_mode), _p(rays_c),
                                                     _p(fwd), cones, *tail, _p(records), _stream())
            else:
                rc = lib.voge_fragments_fwd_iso(_p(p0_c), _p(p1_c), _p(rays_c), _p(fwd), cones, *tail, _p(records), _stream())
        _lib.check(rc, "voge_fragments_fwd (trace only)")
        ctx.save_for_backward(p0_c, p1_c, rays_c, sel_len)
        ctx.sel_idx, ctx.cnt, ctx.records = sel_idx, cnt, records
        ctx.meta = (int(mode), int(sigma_mode), bool(shared), B, N)
        _tag_index(sel_idx, cnt, B * N)
        ctx.mark_non_differentiable(sel_idx, cnt, records)
        ctx.set_materialize_grads(False)
        return sel_idx, sel_len, cnt, records

    @staticmethod
    def backward(ctx, _g_idx, g_len, _g_cnt, _g_rec):
        if g_len is None:
            return (None,) * 9
        lib = _lib.load()
        p0, p1, rays, ln = ctx.saved_tensors
        mode, sigma_mode, shared, B, N = ctx.meta
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[3]:
            raise _lib.VogeHipError("deferred-composite fragments carry no gradient for the rays / the camera centre; "
                                    "VOGE_LAZY_COMPOSITE=0 (or rays that require grad at render time) selects the eager chain")
        sel_idx, cnt = ctx.sel_idx, ctx.cnt
        _, H, W, K = sel_idx.shape
        gh = _dev(g_len, torch.float32, "grad_hit_length")
        g0, g1 = torch.empty_like(p0), torch.empty_like(p1)
        with _on(rays.device):
            nbytes = lib.voge_fragment_bwd_workspace_bytes(B * N)
            ws = _workspace(rays.device, nbytes)
            # (g_weight = NULL: u = 0 everywhere, the `weight` operand is read but multiplies zero -- len stands in)
            if mode == 0:
                rc = lib.voge_fragment_bwd(_p(p0), _p(p1), _p(rays), _p(sel_idx), _p(cnt), _p(ln), None, _p(ln), None, None, 0, 0,
                                           _p(gh), 1.0, B * N, B * H, W, K, _p(ws), nbytes, _p(g0), _p(g1), _stream())
            else:
                rc = lib.voge_fragment_bwd_iso(_p(ctx.records), _p(p1), int(shared), sigma_mode, _p(rays), _p(sel_idx), _p(cnt), _p(ln),
                                               None, _p(ln), None, None, 0, 0, _p(gh), 1.0, B, N, B * H, W, K, _p(ws), nbytes, _p(g0),
                                               _p(g1), _stream())
        _lib.check(rc, "voge_fragment_bwd")
        return None, g0, g1, None, None, None, None, None, None


def _plain(t_, dtype, name):
    """_dev for an input the call only READS through its pointer: no torch op may see it (a `.contiguous()` on a parameter would
    put a node into the graph of a call that has none)."""
    if not t_.is_cuda:
        raise _lib.VogeHipError(f"{name} is on {t_.device}: the VoGE hot path runs on a HIP device only (no CPU fallback)")
    if t_.dtype != dtype or not t_.is_contiguous():
        t_ = t_.detach().to(dtype).contiguous()
    return t_


def frame_trace(verts, sigmas, R, T, focal, pp, band, W, behind, thr_act, n_assign, sigma_mode, occ, origin_out=None):
    """Round 6: the renderer's forward up to the sweep with the CAMERA as input (voge_frame_trace_fwd_iso): no ray-generation
    launch and no autograd node at all -- the trace's three kernels make rays, cones, camera centre and view axis from
    (R, T, focal, pp) themselves, with the ray kernel's own operations, and the sweep leaves the bundle behind for the composite
    and the backward.  verts [N,3] | [B,N,3], sigmas [N] | [B,N], R [B,3,3], T [B,3], focal [B,2], pp [B,2], band = (row0, h,
    stripe_h, pitch).  Fixed cameras only (callers that optimise the camera take pixel_rays + the rays-taking entries).
    origin_out: None | a contiguous fp32 [B,3] tensor that receives the camera centres.
    -> (sel_idx, sel_len, LazyComposite): sel_len carries NO grad_fn here; whoever reads Fragments.vert_hit_length gets a
    differentiable alias on demand (LazyComposite.hit_length, _HitLength) -- the weights' consumers never pay for that node."""
    lib = _lib.load()
    v_c, s_c = _plain(verts, torch.float32, "verts"), _plain(sigmas, torch.float32, "sigmas")
    R_c, T_c = _plain(R, torch.float32, "R"), _plain(T, torch.float32, "T")
    f_c, p_c = _plain(focal, torch.float32, "focal_length"), _plain(pp, torch.float32, "principal_point")
    B = R_c.shape[0]
    row0, h, stripe_h, pitch = band
    shared = v_c.dim() == 2
    assert R_c.shape == (B, 3, 3) and T_c.shape == (B, 3) and f_c.shape == (B, 2) and p_c.shape == (B, 2)
    assert v_c.shape[-1] == 3 and (shared or v_c.shape[0] == B) and s_c.shape == v_c.shape[:-1]
    N, K, dev = v_c.shape[-2], int(n_assign), v_c.device
    empty = torch.empty
    sel_idx = empty((B, h, W, K), dtype=torch.int32, device=dev)
    sel_len = empty((B, h, W, K), dtype=torch.float32, device=dev)
    cnt = empty((B, h, W), dtype=torch.int32, device=dev)
    records = empty((B * N, 4), dtype=torch.float32, device=dev)
    rays = empty((B, h, W, 3), dtype=torch.float32, device=dev)
    with _on(dev):
        nbytes = lib.voge_trace_workspace_bytes(B, N, h, W)
        ws = _workspace(dev, nbytes)
        rc = lib.voge_frame_trace_fwd_iso(v_c.data_ptr(), s_c.data_ptr(), int(shared), int(sigma_mode), R_c.data_ptr(), T_c.data_ptr(),
                                          f_c.data_ptr(), p_c.data_ptr(), int(row0), int(stripe_h), int(pitch), int(bool(behind)), B, N,
                                          int(h), int(W), K, float(thr_act), ws.data_ptr(), nbytes, sel_idx.data_ptr(),
                                          sel_len.data_ptr(), cnt.data_ptr(), records.data_ptr(), rays.data_ptr(), _p(origin_out), _stream())
    if rc:
        _lib.check(rc, "voge_frame_trace_fwd_iso")
    _tag_index(sel_idx, cnt, B * N)
    lz = LazyComposite.__new__(LazyComposite)
    (lz.mode, lz.sigma_mode, lz.shared, lz.occ, lz.B, lz.N, lz.K, lz.p0, lz.p1, lz.sel_idx, lz.sel_len, lz.cnt, lz.records, lz.rays,
     lz.rays_version, lz.grad_mode, lz.idx_version, lz.p0_version, lz.frame, lz.hit_len, lz.gen) = (
        2, int(sigma_mode), shared, float(occ), B, N, K, verts, sigmas, sel_idx, sel_len, cnt, records, rays, 0, torch.is_grad_enabled(),
        sel_idx._version, verts._version, True, None, None)
    return sel_idx, sel_len, lz


def _sigma_kind(sigmas):
    """(kind, shared) of a general sigma tensor: kind 1 = per-axis [.., N, 3], kind 2 = [.., N, 3, 3] (_GeneralPreamble's rule)."""
    kind = 2 if tuple(sigmas.shape[-2:]) == (3, 3) and sigmas.dim() >= 3 else 1
    return kind, sigmas.dim() == (3 if kind == 2 else 2)


def frame_trace_gen(verts, sigmas, R, T, focal, pp, band, W, behind, thr_act, n_assign, occ, origin_out=None):
    """frame_trace for (N,3) / (N,3,3) sigmas (voge_frame_trace_fwd_gen): the camera AND the user's own arrays go in -- the record
    pass centres the vertices and expands 2 * sigma itself (Renderer.py:130-137: no ray launch, no preamble launch, no autograd
    node).  verts [N,3] | [B,N,3]; sigmas [N,3] | [N,3,3] | [B,N,3] | [B,N,3,3].  -> (sel_idx, sel_len, LazyComposite) with
    mode 0, lz.gen = (kind, shared_verts, shared_sigmas) and lz.p0 / lz.p1 the USER's tensors: every backward route of such
    fragments goes through voge_frame_bwd_gen, which hands the gradients to exactly those."""
    lib = _lib.load()
    v_c, s_c = _plain(verts, torch.float32, "verts"), _plain(sigmas, torch.float32, "sigmas")
    R_c, T_c = _plain(R, torch.float32, "R"), _plain(T, torch.float32, "T")
    f_c, p_c = _plain(focal, torch.float32, "focal_length"), _plain(pp, torch.float32, "principal_point")
    B = R_c.shape[0]
    row0, h, stripe_h, pitch = band
    shared_v = v_c.dim() == 2
    kind, shared_s = _sigma_kind(s_c)
    N, K, dev = v_c.shape[-2], int(n_assign), v_c.device
    assert R_c.shape == (B, 3, 3) and T_c.shape == (B, 3) and f_c.shape == (B, 2) and p_c.shape == (B, 2)
    assert v_c.shape[-1] == 3 and (shared_v or v_c.shape[0] == B)
    assert s_c.shape[-(kind + 1)] == N and s_c.shape[-1] == 3 and (shared_s or s_c.shape[0] == B)
    empty = torch.empty
    sel_idx = empty((B, h, W, K), dtype=torch.int32, device=dev)
    sel_len = empty((B, h, W, K), dtype=torch.float32, device=dev)
    cnt = empty((B, h, W), dtype=torch.int32, device=dev)
    records = empty((B * N, 8 if kind == 1 else 12), dtype=torch.float32, device=dev)      # (kind 1: the compact per-axis records)
    rays = empty((B, h, W, 3), dtype=torch.float32, device=dev)
    with _on(dev):
        nbytes = lib.voge_trace_workspace_bytes(B, N, h, W)
        ws = _workspace(dev, nbytes)
        rc = lib.voge_frame_trace_fwd_gen(v_c.data_ptr(), s_c.data_ptr(), int(shared_v), int(shared_s), kind, R_c.data_ptr(), T_c.data_ptr(),
                                          f_c.data_ptr(), p_c.data_ptr(), int(row0), int(stripe_h), int(pitch), int(bool(behind)), B, N,
                                          int(h), int(W), K, float(thr_act), ws.data_ptr(), nbytes, sel_idx.data_ptr(),
                                          sel_len.data_ptr(), cnt.data_ptr(), records.data_ptr(), rays.data_ptr(), _p(origin_out), _stream())
    if rc:
        _lib.check(rc, "voge_frame_trace_fwd_gen")
    _tag_index(sel_idx, cnt, B * N)
    lz = LazyComposite(mode=0, sigma_mode=0, shared=False, occ=float(occ), B=B, N=N, K=K, p0=verts, p1=sigmas, sel_idx=sel_idx,
                       sel_len=sel_len, cnt=cnt, records=records, rays=rays, rays_version=0, grad_mode=torch.is_grad_enabled(),
                       idx_version=sel_idx._version, p0_version=verts._version, frame=True, gen=(kind, int(shared_v), int(shared_s)))
    return sel_idx, sel_len, lz


def _frame_gen_bwd(lib, lz, form, attr, weight, ad, ln, rgb, wsum, bg, thr, g, gs0, gs1, g_hitlen, acc=None):
    """voge_frame_bwd_gen on fragments of frame_trace_gen: -> (g_p0, g_p1, g_attr | None), the gradients of the user's verts / sigmas
    (and attributes).  acc: the accumulator the composite zeroed on its way (good for one call), or None (scratch, filled here)."""
    lz.check()
    kind, shared_v, shared_s = lz.gen
    dev = ln.device
    B, H, W = lz.cnt.shape
    K, P = lz.K, lz.B * lz.N
    g0 = torch.empty(lz.p0.shape, dtype=torch.float32, device=dev)
    g1 = torch.empty(lz.p1.shape, dtype=torch.float32, device=dev)
    g_attr = None if attr is None else torch.empty_like(attr)
    C, Nattr = (0, 0) if attr is None else (attr.shape[1], attr.shape[0])
    with _on(dev):
        zeroed = acc is not None
        if acc is None:
            nbytes = lib.voge_frame_bwd_gen_acc_bytes(P)
            acc = _workspace(dev, nbytes)
        else:
            nbytes = acc.numel()
        rc = lib.voge_frame_bwd_gen(form, _p(lz.records), shared_v, shared_s, kind, _p(lz.rays), _p(attr), _p(lz.sel_idx), _p(lz.cnt),
                                    _p(weight), _p(ad[0]), _p(ln), _p(ad[1]), _p(rgb), _p(wsum), _p(bg), float(thr), _p(g), gs0, gs1,
                                    _p(g_hitlen), lz.occ, lz.B, lz.N, B * H, W, K, C, Nattr, _p(acc), nbytes, int(zeroed), _p(g0), _p(g1),
                                    _p(g_attr), _stream())
    _lib.check(rc, "voge_frame_bwd_gen")
    return g0, g1, g_attr


class _HitLength(torch.autograd.Function):
    """Fragments.vert_hit_length of the camera-input trace as a differentiable tensor (the reference's is: the trace's len itself,
    Aggregation.py:107): forward(verts, sigmas, sel_len, lz) -> an alias of sel_len with this node behind it; backward: the
    trace's chain rule for a gradient of len alone (voge_fragment_bwd_iso with g_weight = NULL), as _TraceLean.backward."""

    @staticmethod
    def forward(ctx, p0, p1, sel_len, lz):
        ctx.lz = lz
        ctx.save_for_backward(p1)
        ctx.set_materialize_grads(False)
        return sel_len.view_as(sel_len)

    @staticmethod
    def backward(ctx, g_len):
        if g_len is None:
            return None, None, None, None
        lib = _lib.load()
        lz = ctx.lz
        lz.check()
        if lz.gen is not None:      # (general forms: the weights' form with g_weight = NULL -- `weight` is read but multiplies zero)
            gh = _dev(g_len, torch.float32, "grad_hit_length")
            g0, g1, _ = _frame_gen_bwd(lib, lz, 2, None, lz.sel_len, (None, None), lz.sel_len, None, None, None, -1.0, None, 0, 0, gh)
            return g0, g1, None, None
        (p1,) = ctx.saved_tensors
        p1 = _dev(p1, torch.float32, "sigmas")
        ln, sel_idx = lz.sel_len, lz.sel_idx
        B, H, W, K = sel_idx.shape
        gh = _dev(g_len, torch.float32, "grad_hit_length")
        g0 = torch.empty(lz.p0.shape, dtype=torch.float32, device=ln.device)
        g1 = torch.empty(lz.p1.shape, dtype=torch.float32, device=ln.device)
        with _on(ln.device):
            nbytes = lib.voge_fragment_bwd_workspace_bytes(lz.B * lz.N)
            ws = _workspace(ln.device, nbytes)
            rc = lib.voge_fragment_bwd_iso(_p(lz.records), _p(p1), int(lz.shared), lz.sigma_mode, _p(lz.rays), _p(sel_idx), _p(lz.cnt),
                                           _p(ln), None, _p(ln), None, None, 0, 0, _p(gh), 1.0, lz.B, lz.N, B * H, W, K, _p(ws), nbytes,
                                           _p(g0), _p(g1), _stream())
        _lib.check(rc, "voge_fragment_bwd")
        return g0, g1, None, None


# VOGE_FRAME_PATH=0 (or ops.FRAME_PATH = False): the renderer generates its ray bundle with voge_rays_fwd and traces through the
# rays-taking entries (round 5's chain); VOGE_FRAME_DIRECT_BWD=0: the frame's backward fills its accumulator itself.  Same results.
FRAME_PATH = os.environ.get("VOGE_FRAME_PATH", "1") != "0"
FRAME_DIRECT_BWD = os.environ.get("VOGE_FRAME_DIRECT_BWD", "1") != "0"


def frame_eligible(verts, sigmas, R, T, focal, pp, n_assign, numel):
    """The camera-input frame path: scalar sigmas on the device, fixed cameras, a non-empty band, 32-bit record offsets."""
    if not (LAZY_COMPOSITE and FRAME_PATH) or n_assign > _lib.MAX_K or os.environ.get("VOGE_FRAGMENTS_KEEP_ACT_DSD", "0") == "1":
        return False
    if not (verts.is_cuda and sigmas.is_cuda) or numel == 0 or verts.shape[-2] == 0:
        return False
    if sigmas.dim() != verts.dim() - 1 and not (sigmas.dim() >= 2 and sigmas.shape[-1] == 3 and sigmas.dtype == torch.float32):
        return False      # (scalar sigmas [N] | [B,N], or the general forms [.., N, 3] / [.., N, 3, 3])
    dev = verts.device
    if R.device != dev or T.device != dev or focal.device != dev or pp.device != dev:      # (fixed cameras: cameras.camera_tensors)
        return False
    return R.shape[0] * verts.shape[-2] < (1 << 26)


class LazyComposite:
    """What the deferred composite needs from a _TraceLean call (nothing in it has a grad_fn except sel_len, which the
    composite nodes take as an input)."""
    __slots__ = ("mode", "sigma_mode", "shared", "occ", "B", "N", "K", "p0", "p1", "sel_idx", "sel_len", "cnt", "records", "rays",
                 "rays_version", "grad_mode", "idx_version", "p0_version", "frame", "hit_len", "gen")

    def __init__(self, **kw):
        self.frame = False      # (made by frame_trace: the shade stage may take the frame entries of ABI 7)
        self.hit_len = None     # (unused; kept for pickles of round 6's first builds)
        self.gen = None         # (frame_trace_gen: (kind, shared_verts, shared_sigmas) -- p0 / p1 are the USER's verts / sigmas)
        for k, v in kw.items():
            setattr(self, k, v)

    def hit_length(self):
        """Fragments.vert_hit_length: sel_len itself, or (frame_trace, whose sel_len has no grad_fn) its differentiable alias,
        made on first request under the render call's autograd mode."""
        if not self.frame:
            return self.sel_len
        # (NOT remembered here: the alias' grad_fn holds this object -- the Fragments that asked keeps what it gets)
        if self.grad_mode and (self.p0.requires_grad or self.p1.requires_grad):
            with torch.enable_grad():
                return _HitLength.apply(self.p0, self.p1, self.sel_len, self)
        return self.sel_len

    def means(self):
        """(general forms) the centres as the trace saw them: contiguous fp32, no grad_fn."""
        assert self.gen is None, "fragments of frame_trace_gen keep the USER's vertices: their backward reads the records"
        return _dev(self.p0.detach(), torch.float32, "means")

    def usable(self):
        """The sweep's outputs must still be what it wrote (nobody edited them through torch in between)."""
        return (self.sel_len._version == 0 and self.rays._version == self.rays_version
                and hit_count_of(self.sel_idx) is not None)

    def check(self):
        """Backward-time guard: what the kernels read outside save_for_backward must be what the forward saw."""
        if self.rays._version != self.rays_version:
            raise RuntimeError("the rays were modified in place after the fragments were traced: the backward would read the "
                               "changed values (as autograd's own check for saved tensors)")
        if self.sel_idx._version != self.idx_version:
            raise RuntimeError("vert_index was modified in place after the fragments were traced: the backward would read the "
                               "changed values (as autograd's own check for saved tensors)")
        if self.mode == 0 and self.p0._version != self.p0_version:
            raise RuntimeError("the Gaussians' centres were modified in place after the fragments were traced (an optimizer "
                               "step between forward and backward?): the backward would read the changed values")

    def grad(self):
        """The autograd mode the RENDER ran under, for whoever composites these fragments later: the reference computes the
        weights inside the renderer call, so reading them first under torch.no_grad() (a feature-bank update, logging) must
        neither drop the graph of a later loss nor build one for fragments rendered without."""
        return torch.set_grad_enabled(self.grad_mode)

    def through(self, weight):
        """Tag freshly composited weights the way fragments() does (a later to_colored_background takes _ShadeThrough).  The
        bookkeeping is filled in when somebody first LOOKS at it (_Through): the frame that never does pays for no detach."""
        # (fragments of frame_trace_gen carry the user's own verts / sigmas: _ShadeThrough's entry points want the centred means
        #  and the expanded forms -- such weights stay untagged and a later to_colored_background takes the ordinary nodes)
        if (self.p0.requires_grad or self.p1.requires_grad) and self.grad_mode and self.gen is None:
            weight.voge_through = _Through(self, weight)
        return weight


class _Through(dict):
    """The `voge_through` bookkeeping of deferred-composite weights (see fragments()), materialised on first lookup.  The version
    counters are read when the object is made -- that is what "untouched since" refers to."""

    def __init__(self, lz, weight):
        dict.__init__(self)
        # (this object hangs on `weight` itself: it must not hold the tensor, or every frame's buffers would wait for the cyclic
        #  collector -- and the allocator would hipMalloc new ones meanwhile: 0.3 ms of host per frame when it happened)
        self._src = (lz, getattr(weight, "voge_act_dsd", (None, None)),      # (general forms: what the composite kept)
                     (weight._version, lz.sel_len._version, lz.rays._version, lz.sel_idx._version))

    def _fill(self):
        lz, (act, dsd), versions = self._src
        self._src = None
        len_d = lz.sel_len.detach()
        self.update(mode=lz.mode, sigma_mode=lz.sigma_mode, shared=lz.shared, occ=lz.occ, B=lz.B, N=lz.N, records=lz.records,
                    rays=lz.rays, act=act, dsd=dsd, len=len_d, cnt=lz.cnt, idx=lz.sel_idx, sigmas=lz.p1.detach(), means=lz.p0.detach(),
                    p0=lz.p0, p1=lz.p1, rays_requires_grad=False, versions=versions)

    def __missing__(self, key):
        if self._src is None:
            raise KeyError(key)
        self._fill()
        return dict.__getitem__(self, key)


def _general_act_dsd(lz, sel_len, need):
    """(act, dsd) buffers the general composite fills for its backward (VOGE_GENERAL_KEEP_ACT_DSD=0: none -- the backward
    re-derives them from the packed records, 19 us slower at the cfg3 size); scalar-sigma fragments never keep them."""
    if lz.mode != 0 or os.environ.get("VOGE_GENERAL_KEEP_ACT_DSD", "1") == "0" or not need or (lz.gen is not None and lz.gen[0] == 1):
        return (None, None)      # (per-axis sigmas on the frame path: re-derived from their 32-byte records, three coefficients)
    return (torch.empty_like(sel_len), torch.empty_like(sel_len))


def _lazy_fragment_bwd(lib, lz, p1, weight, ln, g_weight, K, ad=(None, None)):
    """composite + trace backward in one pass for a deferred composite: -> (g_p0, g_p1)."""
    if lz.gen is not None:
        gw, gs_pix, gs_k = _grad_weight_layout(g_weight, K)
        g0, g1, _ = _frame_gen_bwd(lib, lz, 2, None, weight, ad, ln, None, None, None, -1.0, gw, gs_pix, gs_k, None)
        return g0, g1
    lz.check()
    gw, gs_pix, gs_k = _grad_weight_layout(g_weight, K)
    g0 = torch.empty(lz.p0.shape, dtype=torch.float32, device=ln.device)
    g1 = torch.empty(lz.p1.shape, dtype=torch.float32, device=ln.device)
    B, H, W = lz.cnt.shape
    with _on(ln.device):
        nbytes = lib.voge_fragment_bwd_workspace_bytes(lz.B * lz.N)
        ws = _workspace(ln.device, nbytes)
        if lz.mode == 0:
            rc = lib.voge_fragment_bwd(_p(lz.means()), _p(p1), _p(lz.rays), _p(lz.sel_idx), _p(lz.cnt), _p(weight), _p(ad[0]), _p(ln),
                                       _p(ad[1]), _p(gw), gs_pix, gs_k, None, lz.occ, lz.B * lz.N, B * H, W, K, _p(ws), nbytes, _p(g0),
                                       _p(g1), _stream())
        else:
            rc = lib.voge_fragment_bwd_iso(_p(lz.records), _p(p1), int(lz.shared), lz.sigma_mode, _p(lz.rays), _p(lz.sel_idx), _p(lz.cnt),
                                           _p(weight), None, _p(ln), None, _p(gw), gs_pix, gs_k, None, lz.occ, lz.B, lz.N, B * H, W, K,
                                           _p(ws), nbytes, _p(g0), _p(g1), _stream())
    _lib.check(rc, "voge_fragment_bwd")
    return g0, g1


class _CompositeLean(torch.autograd.Function):
    """aggregation (VoGE/Aggregation.py:82-107) of a _TraceLean call, when the weights are first asked for:
    forward(p0, p1, sel_len, lz) -> weight, valid_num.  The Gaussians are inputs so that the backward (composite + trace,
    one pass: voge_fragment_bwd_iso) can hand them the gradient directly; sel_len gets none from here."""

    @staticmethod
    def forward(ctx, p0, p1, sel_len, lz):
        lib = _lib.load()
        idx, K = lz.sel_idx, lz.K
        weight = torch.empty_like(sel_len)
        valid = torch.empty(idx.shape[:-1], dtype=torch.int64, device=idx.device)
        with _on(idx.device):
            ctx.ad = _general_act_dsd(lz, sel_len, any(ctx.needs_input_grad))
            if lz.gen is not None:
                rc = lib.voge_frame_shade_fwd_rec(lz.gen[0], _p(idx), _p(lz.cnt), _p(sel_len), _p(lz.records), _p(lz.rays), lz.occ, None, None,
                                                  -1.0, idx.numel() // K, K, 0, 0, _p(weight), _p(valid), None, None, None, None,
                                                  _p(ctx.ad[0]), _p(ctx.ad[1]), None, 0, _stream())
            elif lz.mode == 0:
                rc = lib.voge_composite_fwd_rec(_p(idx), _p(lz.cnt), _p(sel_len), _p(lz.records), _p(lz.rays), lz.occ, idx.numel() // K,
                                                K, _p(weight), _p(valid), _p(ctx.ad[0]), _p(ctx.ad[1]), _stream())
            else:
                rc = lib.voge_composite_fwd_iso(_p(idx), _p(lz.cnt), _p(sel_len), _p(lz.records), _p(lz.rays), lz.occ, idx.numel() // K,
                                                K, _p(weight), _p(valid), _stream())
        _lib.check(rc, "voge_composite_fwd (from the records)")
        weight.voge_act_dsd = ctx.ad
        ctx.save_for_backward(_dev(p1, torch.float32, "sigmas"), sel_len, weight)
        ctx.lz = lz
        ctx.mark_non_differentiable(valid)
        ctx.set_materialize_grads(False)
        return weight, valid

    @staticmethod
    def backward(ctx, g_weight, _g_valid):
        if g_weight is None:
            return None, None, None, None
        p1, ln, weight = ctx.saved_tensors
        g0, g1 = _lazy_fragment_bwd(_lib.load(), ctx.lz, p1, weight, ln, g_weight, ctx.lz.K, ctx.ad)
        return g0, g1, None, None


class _CompositeShade(torch.autograd.Function):
    """aggregation + merge_final + get_silhouette + to_colored_background in ONE forward pass
    (voge_composite_shade_fwd_iso): forward(attr, p0, p1, sel_len, lz, bg, thr) -> image, weight, valid_num.
    backward: the image's gradient through voge_fragment_shade_bwd_iso (shade + composite + trace, one kernel); a gradient
    that reaches the weights from elsewhere (a silhouette loss on the same fragments) through voge_fragment_bwd_iso."""

    @staticmethod
    def forward(ctx, attr, p0, p1, sel_len, lz, bg, thr):
        lib = _lib.load()
        attr_c = _dev(attr, torch.float32, "colors")
        bg_c = _dev(bg, torch.float32, "background_color")
        idx, K = lz.sel_idx, lz.K
        Nattr, C = attr_c.shape
        assert bg_c.numel() == C
        check_index_range(idx, Nattr)
        weight = torch.empty_like(sel_len)
        valid = torch.empty(idx.shape[:-1], dtype=torch.int64, device=idx.device)
        rgb = torch.empty(idx.shape[:-1] + (C,), dtype=torch.float32, device=idx.device)
        img = torch.empty_like(rgb)
        wsum = torch.empty(idx.shape[:-1], dtype=torch.float32, device=idx.device)
        # (round 6) fragments of the camera-input trace: the accumulator of the coming backward is allocated here and this launch
        # zeroes it on its way -- no fill launch in front of that backward (voge_frame_shade_bwd_iso)
        ctx.gbuf = None
        if lz.frame and any(ctx.needs_input_grad[:3]) and K <= 128 and FRAME_DIRECT_BWD:
            ctx.gbuf = torch.empty((lz.B * lz.N * (32 if lz.gen is None else (48 if lz.gen[0] == 1 else 64)),), dtype=torch.uint8, device=idx.device)
        with _on(idx.device):
            ctx.ad = _general_act_dsd(lz, sel_len, any(ctx.needs_input_grad))
            args = (_p(idx), _p(lz.cnt), _p(sel_len), _p(lz.records), _p(lz.rays), lz.occ, _p(attr_c), _p(bg_c), float(thr),
                    idx.numel() // K, K, C, Nattr, _p(weight), _p(valid), _p(rgb), _p(img), _p(wsum))
            if lz.gen is not None:
                rc = lib.voge_frame_shade_fwd_rec(lz.gen[0], *args, None, _p(ctx.ad[0]), _p(ctx.ad[1]), _p(ctx.gbuf),
                                                  0 if ctx.gbuf is None else ctx.gbuf.numel(), _stream())
            elif lz.mode == 0:
                rc = lib.voge_composite_shade_fwd_rec(*args, _p(ctx.ad[0]), _p(ctx.ad[1]), _stream())
            elif ctx.gbuf is not None:
                rc = lib.voge_frame_shade_fwd_iso(*args, None, _p(ctx.gbuf), ctx.gbuf.numel(), _stream())
            else:
                rc = lib.voge_composite_shade_fwd_iso(*args, _stream())
        _lib.check(rc, "voge_composite_shade_fwd")
        weight.voge_act_dsd = ctx.ad
        ctx.save_for_backward(attr_c, _dev(p1, torch.float32, "sigmas"), sel_len, weight, rgb, bg_c, wsum)
        ctx.lz, ctx.thr = lz, float(thr)
        ctx.mark_non_differentiable(valid)
        ctx.set_materialize_grads(False)
        return img, weight, valid

    @staticmethod
    def backward(ctx, g_img, g_weight, _g_valid):
        lib = _lib.load()
        attr, p1, ln, weight, rgb, bg, wsum = ctx.saved_tensors
        lz = ctx.lz
        lz.check()
        idx = lz.sel_idx
        B, H, W, K = idx.shape
        Nattr, C = attr.shape
        g_attr = g0 = g1 = None
        if g_img is not None:
            if g_img.dtype == torch.float32 and g_img.is_cuda and all(s == 0 for s in g_img.stride()):
                go, gs_pix, gs_c = g_img, 0, 0      # autograd's broadcast scalar (sum / mean losses), read in place
            else:
                go, gs_pix, gs_c = _dev(g_img, torch.float32, "grad_image"), C, 1
            gbuf, ctx.gbuf = ctx.gbuf, None      # (zeroed by the forward, good for ONE backward: a second one takes the scratch form)
            if lz.gen is not None:      # (general forms of the frame path: the user's verts / sigmas get their gradients directly)
                g0, g1, g_attr = _frame_gen_bwd(lib, lz, 0, attr, weight, ctx.ad, ln, rgb, wsum, bg, ctx.thr, go, gs_pix, gs_c, None, acc=gbuf)
            elif gbuf is not None:
                g_attr = torch.empty_like(attr)
                g0 = torch.empty(lz.p0.shape, dtype=torch.float32, device=idx.device)
                g1 = torch.empty(lz.p1.shape, dtype=torch.float32, device=idx.device)
                with _on(idx.device):
                    rc = lib.voge_frame_shade_bwd_iso(
                        _p(lz.records), _p(p1), int(lz.shared), lz.sigma_mode, _p(lz.rays), _p(attr), _p(idx), _p(lz.cnt), _p(weight),
                        _p(ln), _p(rgb), _p(wsum), _p(bg), ctx.thr, _p(go), gs_pix, gs_c, lz.occ, lz.B, lz.N, B * H, W, K, C, Nattr,
                        _p(gbuf), gbuf.numel(), _p(g0), _p(g1), _p(g_attr), _stream())
                _lib.check(rc, "voge_frame_shade_bwd_iso")
        if g_img is not None and g0 is None:
            g_attr = torch.empty_like(attr)
            g0 = torch.empty(lz.p0.shape, dtype=torch.float32, device=idx.device)
            g1 = torch.empty(lz.p1.shape, dtype=torch.float32, device=idx.device)
            with _on(idx.device):
                nbytes = lib.voge_fragment_bwd_workspace_bytes(lz.B * lz.N)
                ws = _workspace(idx.device, nbytes)
                if lz.mode == 0:
                    rc = lib.voge_fragment_shade_bwd(
                        _p(lz.means()), _p(p1), _p(lz.rays), _p(attr), _p(idx), _p(lz.cnt), _p(weight), _p(ctx.ad[0]), _p(ln),
                        _p(ctx.ad[1]), _p(rgb), _p(wsum), _p(bg), ctx.thr, _p(go), gs_pix, gs_c, lz.occ, lz.B * lz.N, B * H, W, K, C, Nattr,
                        _p(ws), nbytes, _p(g0), _p(g1), _p(g_attr), _stream())
                else:
                    rc = lib.voge_fragment_shade_bwd_iso(
                        _p(lz.records), _p(p1), int(lz.shared), lz.sigma_mode, _p(lz.rays), _p(attr), _p(idx), _p(lz.cnt), _p(weight), None,
                        _p(ln), None, _p(rgb), _p(wsum), _p(bg), ctx.thr, _p(go), gs_pix, gs_c, lz.occ, lz.B, lz.N, B * H, W, K, C, Nattr,
                        _p(ws), nbytes, _p(g0), _p(g1), _p(g_attr), _stream())
            _lib.check(rc, "voge_fragment_shade_bwd")
        if g_weight is not None:
            h0, h1 = _lazy_fragment_bwd(lib, lz, p1, weight, ln, g_weight, K, ctx.ad)
            g0, g1 = (h0, h1) if g0 is None else (g0 + h0, g1 + h1)
        need = ctx.needs_input_grad
        return (g_attr if need[0] else None), (g0 if need[1] else None), (g1 if need[2] else None), None, None, None, None


class _CompositeMerge(torch.autograd.Function):
    """aggregation + merge_final (interpolate_attr) + the per-pixel weight sum in ONE forward pass
    (voge_composite_shade_fwd_iso without a background): forward(attr, p0, p1, sel_len, lz) -> merged attributes,
    weight sum, weight, valid_num.  get_silhouette on the same fragments is min(weight sum, 1), so the reference's training
    pattern -- interpolate_attr + get_silhouette (demo/ShapeFitting.py:217-222) -- is this one kernel forward and ONE
    kernel backward (voge_fragment_merge_bwd_iso: merge backward + the sum's gradient + composite + trace)."""

    @staticmethod
    def forward(ctx, attr, p0, p1, sel_len, lz):
        lib = _lib.load()
        attr_c = _dev(attr, torch.float32, "vert_attr")
        idx, K = lz.sel_idx, lz.K
        Nattr, C = attr_c.shape
        check_index_range(idx, Nattr)
        weight = torch.empty_like(sel_len)
        valid = torch.empty(idx.shape[:-1], dtype=torch.int64, device=idx.device)
        rgb = torch.empty(idx.shape[:-1] + (C,), dtype=torch.float32, device=idx.device)
        wsum = torch.empty(idx.shape[:-1], dtype=torch.float32, device=idx.device)
        # (frame path: the composite writes get_silhouette = min(weight sum, 1) next to the sum, a differentiable OUTPUT of this
        #  node -- the training pattern's silhouette costs no launch forward and none backward: see backward)
        sil = torch.empty_like(wsum) if lz.frame else None
        ctx.gbuf = None      # (see _CompositeShade: the frame path's backward without a fill launch)
        if lz.frame and any(ctx.needs_input_grad[:3]) and K <= 128 and FRAME_DIRECT_BWD:
            ctx.gbuf = torch.empty((lz.B * lz.N * (32 if lz.gen is None else (48 if lz.gen[0] == 1 else 64)),), dtype=torch.uint8, device=idx.device)
        with _on(idx.device):
            ctx.ad = _general_act_dsd(lz, sel_len, any(ctx.needs_input_grad))
            args = (_p(idx), _p(lz.cnt), _p(sel_len), _p(lz.records), _p(lz.rays), lz.occ, _p(attr_c), None, -1.0, idx.numel() // K, K,
                    C, Nattr, _p(weight), _p(valid), _p(rgb), None, _p(wsum))
            nacc = 0 if ctx.gbuf is None else ctx.gbuf.numel()
            if lz.gen is not None:
                rc = lib.voge_frame_shade_fwd_rec(lz.gen[0], *args, _p(sil), _p(ctx.ad[0]), _p(ctx.ad[1]), _p(ctx.gbuf), nacc, _stream())
            elif lz.mode == 0:
                rc = lib.voge_composite_shade_fwd_rec(*args, _p(ctx.ad[0]), _p(ctx.ad[1]), _stream())
            elif lz.frame:
                rc = lib.voge_frame_shade_fwd_iso(*args, _p(sil), _p(ctx.gbuf), nacc, _stream())
            else:
                rc = lib.voge_composite_shade_fwd_iso(*args, _stream())
        _lib.check(rc, "voge_composite_shade_fwd")
        weight.voge_act_dsd = ctx.ad
        ctx.save_for_backward(attr_c, _dev(p1, torch.float32, "sigmas"), sel_len, weight, wsum)
        ctx.lz = lz
        ctx.mark_non_differentiable(valid)
        ctx.set_materialize_grads(False)
        return rgb, wsum, weight, valid, sil

    @staticmethod
    def backward(ctx, g_rgb, g_wsum, g_weight, _g_valid, g_sil=None):
        lib = _lib.load()
        attr, p1, ln, weight, wsum = ctx.saved_tensors
        lz = ctx.lz
        lz.check()
        idx = lz.sel_idx
        B, H, W, K = idx.shape
        Nattr, C = attr.shape
        g_attr = g0 = g1 = None
        # the silhouette's gradient: handed to the fused kernel as it is (with the forward's sums, which say where min(sum, 1)
        # passes it on) when that kernel runs on the forward's accumulator; any other combination turns it into the sum's own
        wsum_fwd = None
        if g_sil is not None:
            if g_wsum is None and (ctx.gbuf is not None or lz.gen is not None):
                g_wsum, wsum_fwd = g_sil, wsum
            else:
                passed = g_sil * torch.where(wsum < 1, 1.0, torch.where(wsum == 1, 0.5, 0.0))
                g_wsum = passed if g_wsum is None else g_wsum + passed
        if g_rgb is not None or g_wsum is not None:
            if g_rgb is None:
                g_rgb = torch.zeros(idx.shape[:-1] + (C,), dtype=torch.float32, device=idx.device)
            if g_rgb.dtype == torch.float32 and g_rgb.is_cuda and all(s == 0 for s in g_rgb.stride()):
                go, gs_pix, gs_c = g_rgb, 0, 0
            else:
                go, gs_pix, gs_c = _dev(g_rgb, torch.float32, "grad_out"), C, 1
            gws = None if g_wsum is None else _dev(g_wsum, torch.float32, "grad_weight_sum")
            gbuf, ctx.gbuf = ctx.gbuf, None      # (zeroed by the forward, good for ONE backward)
            if lz.gen is not None:
                g0, g1, g_attr = _frame_gen_bwd(lib, lz, 1, attr, weight, ctx.ad, ln, wsum_fwd, gws, None, -1.0, go, gs_pix, gs_c, None, acc=gbuf)
            elif gbuf is not None:
                g_attr = torch.empty_like(attr)
                g0 = torch.empty(lz.p0.shape, dtype=torch.float32, device=idx.device)
                g1 = torch.empty(lz.p1.shape, dtype=torch.float32, device=idx.device)
                with _on(idx.device):
                    rc = lib.voge_frame_merge_bwd_iso(
                        _p(lz.records), _p(p1), int(lz.shared), lz.sigma_mode, _p(lz.rays), _p(attr), _p(idx), _p(lz.cnt), _p(weight),
                        _p(ln), _p(go), gs_pix, gs_c, _p(gws), _p(wsum_fwd), lz.occ, lz.B, lz.N, B * H, W, K, C, Nattr, _p(gbuf),
                        gbuf.numel(), _p(g0), _p(g1), _p(g_attr), _stream())
                _lib.check(rc, "voge_frame_merge_bwd_iso")
        if (g_rgb is not None or g_wsum is not None) and g0 is None:
            g_attr = torch.empty_like(attr)
            g0 = torch.empty(lz.p0.shape, dtype=torch.float32, device=idx.device)
            g1 = torch.empty(lz.p1.shape, dtype=torch.float32, device=idx.device)
            with _on(idx.device):
                nbytes = lib.voge_fragment_bwd_workspace_bytes(lz.B * lz.N)
                ws = _workspace(idx.device, nbytes)
                if lz.mode == 0:
                    rc = lib.voge_fragment_merge_bwd(
                        _p(lz.means()), _p(p1), _p(lz.rays), _p(attr), _p(idx), _p(lz.cnt), _p(weight), _p(ctx.ad[0]), _p(ln),
                        _p(ctx.ad[1]), _p(go), gs_pix, gs_c, _p(gws), lz.occ, lz.B * lz.N, B * H, W, K, C, Nattr, _p(ws), nbytes, _p(g0),
                        _p(g1), _p(g_attr), _stream())
                else:
                    rc = lib.voge_fragment_merge_bwd_iso(
                        _p(lz.records), _p(p1), int(lz.shared), lz.sigma_mode, _p(lz.rays), _p(attr), _p(idx), _p(lz.cnt), _p(weight), None,
                        _p(ln), None, _p(go), gs_pix, gs_c, _p(gws), lz.occ, lz.B, lz.N, B * H, W, K, C, Nattr, _p(ws), nbytes, _p(g0),
                        _p(g1), _p(g_attr), _stream())
            _lib.check(rc, "voge_fragment_merge_bwd")
        if g_weight is not None:
            h0, h1 = _lazy_fragment_bwd(lib, lz, p1, weight, ln, g_weight, K, ctx.ad)
            g0, g1 = (h0, h1) if g0 is None else (g0 + h0, g1 + h1)
        need = ctx.needs_input_grad
        return (g_attr if need[0] else None), (g0 if need[1] else None), (g1 if need[2] else None), None, None


def composite_merge(lz, attr):
    """-> merged attributes, weight sum, weight, valid_num; or None when the one-pass form does not apply."""
    K = lz.K
    if K > 128 or attr.dim() != 2 or attr.shape[1] not in (3, 4) or attr.numel() >= (1 << 30) or not lz.usable():
        return None
    if os.environ.get("VOGE_SHADE_THROUGH", "1") == "0":
        return None
    with lz.grad():
        rgb, wsum, weight, valid, sil = _CompositeMerge.apply(attr, lz.p0, lz.p1, lz.sel_len, lz)
    return rgb, wsum, lz.through(weight), valid, sil


def trace_lean(mode, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode=0, occ=1.0):
    """-> (sel_idx, sel_len, LazyComposite): the renderer's forward up to the sweep; the composite is deferred."""
    sel_idx, sel_len, cnt, records = _TraceLean.apply(mode, p0, p1, origin, rays, cam_fwd, thr_act, n_assign, sigma_mode)
    B = rays.shape[0]
    N = p0.shape[-2] if mode == 2 else p0.shape[0] // max(B, 1)
    rays_d = rays.detach()
    lz = LazyComposite(mode=int(mode), sigma_mode=int(sigma_mode), shared=bool(mode == 2 and p0.dim() == 2), occ=float(occ), B=B, N=N,
                       K=int(n_assign), p0=p0, p1=p1, sel_idx=sel_idx, sel_len=sel_len, cnt=cnt, records=records, rays=rays_d,
                       rays_version=rays_d._version, grad_mode=torch.is_grad_enabled(), idx_version=sel_idx._version,
                       p0_version=p0._version)
    return sel_idx, sel_len, lz


def lazy_eligible(mode, p0, p1, origin, rays, n_assign):
    """Deferred composite: record gathers in 32-bit offsets, nobody differentiating the rays.  (mode 0 = full 3x3 forms,
    VOGE_LAZY_GENERAL=0 keeps those on the eager chain with act / dsd.)"""
    if not LAZY_COMPOSITE or os.environ.get("VOGE_FRAGMENTS_KEEP_ACT_DSD", "0") == "1":
        return False
    if mode == 0 and os.environ.get("VOGE_LAZY_GENERAL", "1") == "0":
        return False
    B = rays.shape[0]
    P = p0.shape[-2] * B if mode == 2 else p0.shape[0]
    if P >= (1 << 26) or P == 0 or rays.numel() == 0 or rays.requires_grad or (origin is not None and origin.requires_grad):
        return False
    return p0.is_cuda and rays.is_cuda


def composite_lean(lz):
    """-> weight, valid_num of a deferred composite (the weights carry `voge_through` like fragments()' do)."""
    with lz.grad():
        weight, valid = _CompositeLean.apply(lz.p0, lz.p1, lz.sel_len, lz)
    return lz.through(weight), valid


def composite_shade(lz, attr, bg, thr):
    """-> image, weight, valid_num, or None when the one-pass form does not apply (K % 4, channel count, offsets)."""
    K = lz.K
    if K > 128 or attr.dim() != 2 or attr.shape[1] not in (3, 4) or attr.numel() >= (1 << 30) or not lz.usable():
        return None      # (K <= 128: the image's backward is voge_fragment_shade_bwd_iso)
    if os.environ.get("VOGE_SHADE_THROUGH", "1") == "0":
        return None
    if torch.is_grad_enabled() == lz.grad_mode:      # (the usual case: no mode switch to pay for)
        img, weight, valid = _CompositeShade.apply(attr, lz.p0, lz.p1, lz.sel_len, lz, bg, thr)
    else:
        with lz.grad():
            img, weight, valid = _CompositeShade.apply(attr, lz.p0, lz.p1, lz.sel_len, lz, bg, thr)
    return img, lz.through(weight), valid


class _ShadeThrough(torch.autograd.Function):
    """to_colored_background on fragments this renderer made: the forward is _Shade's, the
    backward runs shade -> composite -> trace as ONE kernel (voge_fragment_shade_bwd_iso for scalar sigmas,
    voge_fragment_shade_bwd for full 3x3 forms) and hands the gradients
    straight to the colours AND to the Gaussians' (verts | means, sigmas | a), which are inputs of this node for that
    purpose; the weights get no gradient from here.  Every stage is linear in the gradient of the weights, so whatever
    else consumes the same weights (a silhouette loss, a second image) still flows through _Fragments.backward and the
    contributions add up in the parameters' .grad exactly as with separate nodes."""

    @staticmethod
    def forward(ctx, attr, weight, p0, p1, th, idx, valid_num, bg, thr):
        lib = _lib.load()
        attr_c = _dev(attr, torch.float32, "colors")
        w = _dev(weight, torch.float32, "weight")
        vn = _dev(valid_num, torch.int64, "valid_num")
        bg_c = _dev(bg, torch.float32, "background_color")
        K = idx.shape[-1]
        npix = idx.numel() // max(K, 1)
        Nattr, C = attr_c.shape
        assert C <= 4 and bg_c.numel() == C
        check_index_range(idx, Nattr)
        rgb = torch.empty(idx.shape[:-1] + (C,), dtype=torch.float32, device=idx.device)
        img = torch.empty_like(rgb)
        wsum = torch.empty(idx.shape[:-1], dtype=torch.float32, device=idx.device)
        with _on(idx.device):
            rc = lib.voge_shade_fwd(_p(attr_c), _p(idx), _p(w), _p(vn), _p(bg_c), float(thr), npix, K, C, Nattr, 1,
                                    _p(rgb), _p(img), None, _p(wsum), _stream())
        _lib.check(rc, "voge_shade_fwd")
        ctx.save_for_backward(attr_c, w, rgb, bg_c, wsum)
        ctx.th, ctx.idx, ctx.thr = th, idx, float(thr)
        return img

    @staticmethod
    def backward(ctx, g_img):
        lib = _lib.load()
        attr, w, rgb, bg, wsum = ctx.saved_tensors
        th, idx = ctx.th, ctx.idx
        # what the kernel reads outside save_for_backward must still be what the forward saw
        if (th["len"]._version, th["rays"]._version) != th["versions"][1:3]:
            raise RuntimeError("vert_hit_length or the rays were modified in place after to_colored_background: "
                               "the fused backward would read the changed values (as autograd's own check for saved tensors)")
        B, H, W, K = idx.shape
        Nattr, C = attr.shape
        # the gradient as it comes: a contiguous image, or autograd's broadcast scalar (sum / mean losses) read in place
        if g_img.dtype == torch.float32 and g_img.is_cuda and all(s == 0 for s in g_img.stride()):
            go, gs_pix, gs_c = g_img, 0, 0
        else:
            go, gs_pix, gs_c = _dev(g_img, torch.float32, "grad_image"), C, 1
        g_attr = torch.empty_like(attr)
        p0, p1 = th["p0"], th["p1"]
        g0 = torch.empty(p0.shape, dtype=torch.float32, device=idx.device)
        g1 = torch.empty(p1.shape, dtype=torch.float32, device=idx.device)
        P = th["B"] * th["N"]
        with _on(idx.device):
            nbytes = lib.voge_fragment_bwd_workspace_bytes(P)
            ws = _workspace(idx.device, nbytes)
            if th["mode"] == 0:      # full 3x3 forms: gradients of (mus, isigmas) as given
                rc = lib.voge_fragment_shade_bwd(
                    _p(th["means"]), _p(th["sigmas"]), _p(th["rays"]), _p(attr), _p(idx), _p(th["cnt"]), _p(w), _p(th["act"]),
                    _p(th["len"]), _p(th["dsd"]), _p(rgb), _p(wsum), _p(bg), ctx.thr, _p(go), gs_pix, gs_c, th["occ"], P, B * H, W, K,
                    C, Nattr, _p(ws), nbytes, _p(g0), _p(g1), _p(g_attr), _stream())
            else:
                rc = lib.voge_fragment_shade_bwd_iso(
                    _p(th["records"]), _p(th["sigmas"]), int(th["shared"]), th["sigma_mode"], _p(th["rays"]), _p(attr), _p(idx),
                    _p(th["cnt"]), _p(w), _p(th["act"]), _p(th["len"]), _p(th["dsd"]), _p(rgb), _p(wsum), _p(bg), ctx.thr, _p(go),
                    gs_pix, gs_c, th["occ"], th["B"], th["N"], B * H, W, K, C, Nattr, _p(ws), nbytes, _p(g0), _p(g1), _p(g_attr),
                    _stream())
        _lib.check(rc, "voge_fragment_shade_bwd")
        need = ctx.needs_input_grad
        return (g_attr if need[0] else None), None, (g0 if need[2] else None), (g1 if need[3] else None), None, None, None, None, None


def through_of(fragments_weight, idx):
    """The bookkeeping fragments() left on the weights, if it still describes (fragments_weight, idx): same memory, not
    modified through torch since.  None otherwise."""
    th = getattr(fragments_weight, "voge_through", None)
    if th is None:
        return None
    tidx = th["idx"]
    if idx is not tidx and not (idx.data_ptr() == tidx.data_ptr() and idx.numel() == tidx.numel() and idx.is_contiguous()
                                and idx.dtype == tidx.dtype):
        return None
    v = th["versions"]
    if fragments_weight._version != v[0] or th["len"]._version != v[1] or th["rays"]._version != v[2] or tidx._version != v[3]:
        return None
    return th


def shade_through(attr, fragments_weight, idx, valid_num, bg, thr):
    """The fused-backward form of shade() when `fragments_weight` still is the untouched output of fragments() (K <= 128,
    <= 4 channels, nobody watching the weights' own gradient); None otherwise.  VOGE_SHADE_THROUGH=0 disables it (e.g. to
    take torch.autograd.grad with respect to the weights themselves: this node hands them no gradient)."""
    if not torch.is_grad_enabled() or os.environ.get("VOGE_SHADE_THROUGH", "1") == "0":
        return None
    th = through_of(fragments_weight, idx)
    if th is None:
        return None
    K = idx.shape[-1]
    if K > 128 or attr.dim() != 2 or attr.shape[1] > 4 or th["cnt"] is None:
        return None
    if fragments_weight.retains_grad or fragments_weight._backward_hooks or th["rays_requires_grad"]:
        return None
    if th["B"] * th["N"] >= (1 << 26) or attr.numel() >= (1 << 30):      # 32-bit byte offsets of the kernel's gathers
        return None
    img = _ShadeThrough.apply(attr, fragments_weight, th["p0"], th["p1"], th, th["idx"], valid_num, bg, thr)
    return img if idx is th["idx"] else img.view(idx.shape[:-1] + (attr.shape[1],))


class _Composite(torch.autograd.Function):
    """Fused replacement of get_cross_activation + assign2weight (VoGE/Aggregation.py:30-79)
    and of their autograd backward.  (sel_idx, sel_act, sel_len, sel_dsd, occ) -> weight, valid_num."""

    @staticmethod
    def forward(ctx, sel_idx, sel_act, sel_len, sel_dsd, occ):
        lib = _lib.load()
        idx = _dev(sel_idx, torch.int32, "sel_idx")
        # the trace forward leaves its per-pixel hit count on the index tensor it returns
        # (voge_hit_count): valid_num then needs no pass over idx
        # (ignored when sel_idx was edited through torch since: hit_count_of compares the version counter)
        cnt = hit_count_of(sel_idx) if idx is sel_idx else None
        act = _dev(sel_act, torch.float32, "sel_act")
        ln = _dev(sel_len, torch.float32, "sel_len")
        dsd = _dev(sel_dsd, torch.float32, "sel_dsd")
        K = idx.shape[-1]
        npix = idx.numel() // max(K, 1)
        weight = torch.empty_like(act)
        valid = torch.empty(idx.shape[:-1], dtype=torch.int64, device=idx.device)
        with _on(idx.device):
            rc = lib.voge_composite_fwd(_p(idx), _p(cnt), _p(act), _p(ln), _p(dsd), float(occ), npix, K, _p(weight),
                                        _p(valid), _stream())
        _lib.check(rc, "voge_composite_fwd")
        ctx.save_for_backward(act, ln, dsd, weight)
        ctx.cnt = cnt      # the trace's hit count (or None): lets the backward skip empty slots / pixels
        ctx.occ = float(occ)
        ctx.mark_non_differentiable(valid)
        ctx.set_materialize_grads(False)
        return weight, valid

    @staticmethod
    def backward(ctx, g_weight, _g_valid):
        lib = _lib.load()
        if g_weight is None:
            return None, None, None, None, None
        act, ln, dsd, weight = ctx.saved_tensors
        K = act.shape[-1]
        npix = act.numel() // max(K, 1)
        gw = _dev(g_weight, torch.float32, "grad_weight")
        g_act = torch.empty_like(act)
        g_len = torch.empty_like(act)
        g_dsd = torch.empty_like(act)
        with _on(act.device):
            rc = lib.voge_composite_bwd(_p(act), _p(ln), _p(dsd), _p(weight), _p(ctx.cnt), _p(gw), ctx.occ, npix, K, _p(g_act), _p(g_len),
                                        _p(g_dsd), _stream())
        _lib.check(rc, "voge_composite_bwd")
        return None, g_act, g_len, g_dsd, None


class _Merge(torch.autograd.Function):
    """merge_final (VoGE/Aggregation.py:111-141): (attr [N,C], weight, idx, valid_num) -> [..., C].
    Mutates idx in place (-1 -> 0) exactly like the reference (:131)."""

    @staticmethod
    def forward(ctx, attr, weight, idx, valid_num):
        lib = _lib.load()
        attr_c = _dev(attr, torch.float32, "vert_attr")
        w = _dev(weight, torch.float32, "weight")
        if not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()):
            raise _lib.VogeHipError("vert_index must be a contiguous int32 tensor on the HIP device")
        vn = _dev(valid_num, torch.int64, "valid_num")
        K = idx.shape[-1]
        npix = idx.numel() // max(K, 1)
        Nattr, C = attr_c.shape
        check_index_range(idx, Nattr)
        out = torch.empty(idx.shape[:-1] + (C,), dtype=torch.float32, device=idx.device)
        with _on(idx.device):
            rc = lib.voge_merge_fwd(_p(attr_c), _p(idx), _p(w), _p(vn), npix, K, C, Nattr, 1, _p(out), _stream())
        _lib.check(rc, "voge_merge_fwd")
        # idx is an int tensor outside autograd; it is kept as a plain attribute because the in-place
        # fix (-1 -> 0) may be re-applied by later merges of the same fragments.  The backward masks by
        # valid_num and reads a negative index as 0, so it is insensitive to whether the fix has run.
        ctx.save_for_backward(attr_c, w, vn)
        ctx.idx = idx
        ctx.needs = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        attr, w, vn = ctx.saved_tensors
        idx = ctx.idx
        K = idx.shape[-1]
        npix = idx.numel() // max(K, 1)
        Nattr, C = attr.shape
        go = _dev(g_out, torch.float32, "grad_out")
        g_attr = torch.empty_like(attr) if ctx.needs[0] else None
        g_w = torch.empty_like(w) if ctx.needs[1] else None
        with _on(idx.device):
            Wd = idx.shape[-2] if idx.dim() >= 3 else npix
            rc = lib.voge_merge_bwd(_p(attr), _p(idx), _p(w), _p(vn), _p(go), npix // max(Wd, 1), Wd, K, C, Nattr,
                                    _p(g_attr), _p(g_w), _stream())
        _lib.check(rc, "voge_merge_bwd")
        return g_attr, g_w, None, None


class _Blend(torch.autograd.Function):
    """get_silhouette + to_colored_background (VoGE/Renderer.py:157-171)."""

    @staticmethod
    def forward(ctx, rgb, weight, bg, thr):
        lib = _lib.load()
        rgb_c = _dev(rgb, torch.float32, "rgb")
        w = _dev(weight, torch.float32, "weight")
        bg_c = _dev(bg, torch.float32, "background_color")
        K = w.shape[-1]
        C = rgb_c.shape[-1]
        npix = w.numel() // max(K, 1)
        assert bg_c.numel() == C and rgb_c.numel() == npix * C
        out = torch.empty_like(rgb_c)
        with _on(w.device):
            rc = lib.voge_blend_fwd(_p(rgb_c), _p(w), _p(bg_c), float(thr), npix, K, C, _p(out), None, _stream())
        _lib.check(rc, "voge_blend_fwd")
        ctx.save_for_backward(rgb_c, w, bg_c)
        ctx.thr = float(thr)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        rgb, w, bg = ctx.saved_tensors
        K = w.shape[-1]
        C = rgb.shape[-1]
        npix = w.numel() // max(K, 1)
        go = _dev(g_out, torch.float32, "grad_out")
        g_rgb = torch.empty_like(rgb) if ctx.needs_input_grad[0] else None
        g_w = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        with _on(w.device):
            rc = lib.voge_blend_bwd(_p(rgb), _p(w), _p(bg), ctx.thr, _p(go), npix, K, C, _p(g_rgb), _p(g_w), _stream())
        _lib.check(rc, "voge_blend_bwd")
        return g_rgb, g_w, None, None


class _Shade(torch.autograd.Function):
    """interpolate_attr + get_silhouette + to_colored_background in one kernel each way
    (VoGE/Renderer.py:153-171, VoGE/Aggregation.py:111-141); attr has <= 4 channels."""

    @staticmethod
    def forward(ctx, attr, weight, idx, valid_num, bg, thr):
        lib = _lib.load()
        attr_c = _dev(attr, torch.float32, "colors")
        w = _dev(weight, torch.float32, "weight")
        if not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()):
            raise _lib.VogeHipError("vert_index must be a contiguous int32 tensor on the HIP device")
        vn = _dev(valid_num, torch.int64, "valid_num")
        bg_c = _dev(bg, torch.float32, "background_color")
        K = idx.shape[-1]
        npix = idx.numel() // max(K, 1)
        Nattr, C = attr_c.shape
        assert C <= 4 and bg_c.numel() == C
        check_index_range(idx, Nattr)
        rgb = torch.empty(idx.shape[:-1] + (C,), dtype=torch.float32, device=idx.device)
        img = torch.empty_like(rgb)
        wsum = torch.empty(idx.shape[:-1], dtype=torch.float32, device=idx.device)
        with _on(idx.device):
            rc = lib.voge_shade_fwd(_p(attr_c), _p(idx), _p(w), _p(vn), _p(bg_c), float(thr), npix, K, C, Nattr, 1,
                                    _p(rgb), _p(img), None, _p(wsum), _stream())
        _lib.check(rc, "voge_shade_fwd")
        ctx.save_for_backward(attr_c, w, vn, rgb, bg_c, wsum)
        ctx.idx = idx      # see _Merge: kept outside the version counter on purpose
        ctx.thr = float(thr)
        return img

    @staticmethod
    def backward(ctx, g_img):
        lib = _lib.load()
        attr, w, vn, rgb, bg, wsum = ctx.saved_tensors
        idx = ctx.idx
        K = idx.shape[-1]
        npix = idx.numel() // max(K, 1)
        Nattr, C = attr.shape
        go = _dev(g_img, torch.float32, "grad_image")
        g_attr = torch.empty_like(attr) if ctx.needs_input_grad[0] else None
        g_w = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        Wd = idx.shape[-2] if idx.dim() >= 3 else npix
        with _on(idx.device):
            rc = lib.voge_shade_bwd(_p(attr), _p(idx), _p(w), _p(vn), _p(rgb), _p(wsum), _p(bg), ctx.thr, _p(go),
                                    npix // max(Wd, 1), Wd, K, C, Nattr, _p(g_attr), _p(g_w), _stream())
        _lib.check(rc, "voge_shade_bwd")
        return g_attr, g_w, None, None, None, None


class _Silhouette(torch.autograd.Function):
    """get_silhouette (VoGE/Renderer.py:157-159): min(sum_k w_k, 1).  The backward is one value per pixel
    (g_sil * [sum w < 1]), returned as a [.., K] view with stride 0 along the slots: nothing of size [.., K] is
    written, and voge_fragment_bwd* reads the view in place when the silhouette is the weights' only consumer."""

    @staticmethod
    def forward(ctx, weight):
        lib = _lib.load()
        w = _dev(weight, torch.float32, "weight")
        K = w.shape[-1]
        npix = w.numel() // max(K, 1)
        sil = torch.empty(w.shape[:-1], dtype=torch.float32, device=w.device)
        wsum = torch.empty_like(sil)
        with _on(w.device):
            rc = lib.voge_silhouette_fwd(_p(w), npix, K, _p(sil), _p(wsum), _stream())
        _lib.check(rc, "voge_silhouette_fwd")
        ctx.save_for_backward(wsum)
        ctx.K = K
        return sil

    @staticmethod
    def backward(ctx, g_sil):
        lib = _lib.load()
        (wsum,) = ctx.saved_tensors
        go = _dev(g_sil, torch.float32, "grad_sil")
        g_pix = torch.empty_like(wsum)
        with _on(wsum.device):
            rc = lib.voge_silhouette_bwd(_p(wsum), _p(go), wsum.numel(), _p(g_pix), _stream())
        _lib.check(rc, "voge_silhouette_bwd")
        return g_pix.unsqueeze(-1).expand(*wsum.shape, ctx.K)


class _PixelRays(torch.autograd.Function):
    """Ray bundle of VoGE/Renderer.py:124-130: (R [B,3,3], T [B,3], focal [B,2], pp [B,2]) ->
    unit world-space directions [B,h,W,3] of image rows row0..row0+h-1 and the camera centres [B,3]."""

    @staticmethod
    def forward(ctx, R, T, focal, pp, row0, h, W, stripe_h=None, pitch=0):
        lib = _lib.load()
        stripe_h = max(int(h), 1) if stripe_h is None else int(stripe_h)      # (None: one contiguous band)
        R_c, T_c = _dev(R, torch.float32, "R"), _dev(T, torch.float32, "T")
        f_c, p_c = _dev(focal, torch.float32, "focal_length"), _dev(pp, torch.float32, "principal_point")
        B = R_c.shape[0]
        assert R_c.shape == (B, 3, 3) and T_c.shape == (B, 3) and f_c.shape == (B, 2) and p_c.shape == (B, 2)
        rays = torch.empty((B, h, W, 3), dtype=torch.float32, device=R_c.device)
        origin = torch.empty((B, 3), dtype=torch.float32, device=R_c.device)
        # bounding cones of the band's 32x32-pixel super-tiles: the trace's culling aid, made here for free
        cones = torch.empty((max(int(lib.voge_cones_floats(B, int(h), int(W))), 1),), dtype=torch.float32, device=R_c.device)
        with _on(R_c.device):
            rc = lib.voge_rays_striped_fwd(_p(R_c), _p(T_c), _p(f_c), _p(p_c), B, int(row0), int(h), stripe_h, int(pitch), int(W),
                                           _p(rays), _p(origin), _p(cones), _stream())
        _lib.check(rc, "voge_rays_striped_fwd")
        ctx.save_for_backward(R_c, T_c, f_c, p_c)
        ctx.geom = (int(row0), int(h), int(W), stripe_h, int(pitch))
        ctx.mark_non_differentiable(cones)
        return rays, origin, cones

    @staticmethod
    def backward(ctx, g_rays, g_origin, _g_cones):
        lib = _lib.load()
        R, T, f, pp = ctx.saved_tensors
        row0, h, W, stripe_h, pitch = ctx.geom
        B = R.shape[0]
        gr = None if g_rays is None else _dev(g_rays, torch.float32, "grad_rays")
        go = None if g_origin is None else _dev(g_origin, torch.float32, "grad_origin")
        need = ctx.needs_input_grad
        g_R = torch.empty_like(R) if need[0] else None
        g_T = torch.empty_like(T) if need[1] else None
        g_f = torch.empty_like(f) if need[2] else None
        g_p = torch.empty_like(pp) if need[3] else None
        scratch = torch.empty((B, 16), dtype=torch.float32, device=R.device)
        with _on(R.device):
            rc = lib.voge_rays_striped_bwd(_p(R), _p(T), _p(f), _p(pp), _p(gr), _p(go), B, row0, h, stripe_h, pitch, W, _p(scratch),
                                           _p(g_R), _p(g_T), _p(g_f), _p(g_p), _stream())
        _lib.check(rc, "voge_rays_striped_bwd")
        return g_R, g_T, g_f, g_p, None, None, None, None, None


class _RayTraceVoGERay(torch.autograd.Function):
    """Dense trace, mirror of VoGE/RayTracing.py:209-220: (mus [M,3], sigmas [M,3,3], rays [N,3]) ->
    hit_len, hit_act, hit_dsd [N,M]."""

    @staticmethod
    def forward(ctx, mus, sigmas, rays):
        lib = _lib.load()
        mus_c, sig_c, rays_c = _dev(mus, torch.float32, "mus"), _dev(sigmas, torch.float32, "sigmas"), _dev(rays, torch.float32, "rays")
        M, N = mus_c.shape[0], rays_c.shape[0]
        out = [torch.empty((N, M), dtype=torch.float32, device=rays_c.device) for _ in range(3)]
        with _on(rays_c.device):
            rc = lib.voge_ray_dense_fwd(_p(mus_c), _p(sig_c), _p(rays_c), M, N, _p(out[0]), _p(out[1]), _p(out[2]), _stream())
        _lib.check(rc, "voge_ray_dense_fwd")
        ctx.save_for_backward(mus_c, sig_c, rays_c)
        return tuple(out)

    @staticmethod
    def backward(ctx, g_len, g_act, g_dsd):
        lib = _lib.load()
        mus, sig, rays = ctx.saved_tensors
        M, N = mus.shape[0], rays.shape[0]
        gs = [torch.zeros((N, M), dtype=torch.float32, device=rays.device) if g is None else _dev(g, torch.float32, "grad")
              for g in (g_len, g_act, g_dsd)]
        g_ray, g_mus, g_sig = torch.empty_like(rays), torch.empty_like(mus), torch.empty_like(sig)
        with _on(rays.device):
            rc = lib.voge_ray_dense_bwd(_p(mus), _p(sig), _p(rays), _p(gs[0]), _p(gs[1]), _p(gs[2]), M, N, _p(g_ray),
                                        _p(g_mus), _p(g_sig), _stream())
        _lib.check(rc, "voge_ray_dense_bwd")
        return g_mus, g_sig, g_ray


class _FindNearestK(torch.autograd.Function):
    """Mirror of VoGE/RayTracing.py:223-241 with the intended gradient (zero for empty slots)."""

    @staticmethod
    def forward(ctx, hit_len_in, hit_act_in, hit_dsd_in, thr_act, K):
        lib = _lib.load()
        l, a, d = (_dev(t, torch.float32, "hit") for t in (hit_len_in, hit_act_in, hit_dsd_in))
        N, M = l.shape
        idx = torch.empty((N, K), dtype=torch.int32, device=l.device)
        outs = [torch.empty((N, K), dtype=torch.float32, device=l.device) for _ in range(3)]
        with _on(l.device):
            rc = lib.voge_find_nearest_k(_p(l), _p(a), _p(d), float(thr_act), M, int(K), N, _p(idx), _p(outs[0]),
                                         _p(outs[1]), _p(outs[2]), _stream())
        _lib.check(rc, "voge_find_nearest_k")
        ctx.save_for_backward(idx)
        ctx.M = M
        ctx.mark_non_differentiable(idx)
        return idx, outs[0], outs[1], outs[2]

    @staticmethod
    def backward(ctx, _g_idx, g_len, g_act, g_dsd):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        N, K = idx.shape
        gs = [torch.zeros((N, K), dtype=torch.float32, device=idx.device) if g is None else _dev(g, torch.float32, "grad")
              for g in (g_len, g_act, g_dsd)]
        gi = [torch.empty((N, ctx.M), dtype=torch.float32, device=idx.device) for _ in range(3)]
        with _on(idx.device):
            rc = lib.voge_find_nearest_k_bwd(_p(idx), _p(gs[0]), _p(gs[1]), _p(gs[2]), ctx.M, K, N, _p(gi[0]), _p(gi[1]),
                                             _p(gi[2]), _stream())
        _lib.check(rc, "voge_find_nearest_k_bwd")
        return gi[0], gi[1], gi[2], None, None


class _ScatterAttr(torch.autograd.Function):
    """Transpose of merge_final: out[v, c] = sum over slots (pixel, k < valid) with index v of
    weight * pix_attr[pixel, c].  This is sample_voge (VoGE/csrc/sample_voge/sample_voge.cu:35-66)
    when pix_attr = [image | 1]; it runs on the merge-backward kernel and its own backward on the
    merge forward/backward kernels."""

    @staticmethod
    def forward(ctx, pix_attr, weight, idx, valid_num, n_vert):
        lib = _lib.load()
        pa = _dev(pix_attr, torch.float32, "image")
        w = _dev(weight, torch.float32, "weight")
        ix = _dev(idx, torch.int32, "vert_index")
        vn = _dev(valid_num, torch.int64, "valid_num")
        K = ix.shape[-1]
        C = pa.shape[-1]
        npix = ix.numel() // max(K, 1)
        Wd = ix.shape[-2] if ix.dim() >= 3 else npix
        out = torch.empty((int(n_vert), C), dtype=torch.float32, device=ix.device)
        with _on(ix.device):
            rc = lib.voge_merge_bwd(None, _p(ix), _p(w), _p(vn), _p(pa), npix // max(Wd, 1), Wd, K, C, int(n_vert), _p(out),
                                    None, _stream())
        _lib.check(rc, "voge_merge_bwd")
        ctx.save_for_backward(pa, w, ix, vn)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        pa, w, ix, vn = ctx.saved_tensors
        K = ix.shape[-1]
        C = pa.shape[-1]
        npix = ix.numel() // max(K, 1)
        Wd = ix.shape[-2] if ix.dim() >= 3 else npix
        go = _dev(g_out, torch.float32, "grad_features")
        nv = go.shape[0]
        g_pa = torch.empty_like(pa) if ctx.needs_input_grad[0] else None
        g_w = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        with _on(ix.device):
            if g_pa is not None:   # g_pix_attr = merge(g_out): sum_k w * g_out[idx_k]
                rc = lib.voge_merge_fwd(_p(go), _p(ix), _p(w), _p(vn), npix, K, C, nv, 0, _p(g_pa), _stream())
                _lib.check(rc, "voge_merge_fwd")
            if g_w is not None:    # g_w = <pix_attr, g_out[idx]>
                rc = lib.voge_merge_bwd(_p(go), _p(ix), _p(w), _p(vn), _p(pa), npix // max(Wd, 1), Wd, K, C, nv, None,
                                        _p(g_w), _stream())
                _lib.check(rc, "voge_merge_bwd")
        return g_pa, g_w, None, None, None


def rasterize_points_coarse(points, cloud_to_packed_first_idx, num_points_per_cloud, image_size, radius, bin_size,
                            max_points_per_bin):
    """The reference's coarse stage (VoGE._C.rasterize_points_coarse, rasterize_coarse.h:18-25) -> bin_elems
    [B,BH,BW,M] int32, -1 padded.  No gradient, like the reference (RayTracing.py:126-151)."""
    lib = _lib.load()
    pts = _dev(points.detach(), torch.float32, "points")
    rad = _dev(radius.detach(), torch.float32, "radius")
    first = _dev(cloud_to_packed_first_idx, torch.int64, "cloud_to_packed_first_idx")
    num = _dev(num_points_per_cloud, torch.int64, "num_points_per_cloud")
    assert pts.dim() == 2 and pts.shape[1] == 3 and rad.shape == (pts.shape[0], 2)
    H, W = int(image_size[0]), int(image_size[1])
    B = first.shape[0]
    out = torch.empty((B, 1 + (H - 1) // int(bin_size), 1 + (W - 1) // int(bin_size), int(max_points_per_bin)),
                      dtype=torch.int32, device=pts.device)
    with _on(pts.device):
        rc = lib.voge_bin_gaussians(_p(pts), _p(first), _p(num), B, pts.shape[0], H, W, _p(rad), int(bin_size),
                                    int(max_points_per_bin), _p(out), _stream())
    _lib.check(rc, "voge_bin_gaussians")
    return out


def scatter_attr(pix_attr, weight, idx, valid_num, n_vert):
    return _ScatterAttr.apply(pix_attr, weight, idx, valid_num, n_vert)


def scatter_max(weight, idx, n_vert):
    lib = _lib.load()
    w = _dev(weight.detach(), torch.float32, "weight")
    ix = _dev(idx, torch.int32, "vert_index")
    out = torch.empty((int(n_vert),), dtype=torch.float32, device=w.device)
    with _on(w.device):
        rc = lib.voge_scatter_max(_p(w), _p(ix), w.numel(), int(n_vert), _p(out), _stream())
    _lib.check(rc, "voge_scatter_max")
    return out


def pixel_rays(R, T, focal, pp, row0, h, W, stripe_h=None, pitch=0):
    """rays [B,h,W,3] of image rows row0 .. row0+h-1, or (stripe_h, pitch given) of every pitch-th stripe of stripe_h rows
    from row0 on, h rows in total; origin [B,3]."""
    rays, origin, cones = _PixelRays.apply(R, T, focal, pp, row0, h, W, stripe_h, pitch)
    rays.voge_cones = (cones, rays._version)      # see cones_of()
    return rays, origin


def ray_trace_fine(mus, isigmas, rays, bin_points, thr_act, bin_size, n_assign):
    return _RayTraceVoGE.apply(mus, isigmas, rays, bin_points, thr_act, bin_size, n_assign)


def composite(sel_idx, sel_act, sel_len, sel_dsd, occ=1.0):
    return _Composite.apply(sel_idx, sel_act, sel_len, sel_dsd, occ)


def merge(attr, weight, idx, valid_num):
    return _Merge.apply(attr, weight, idx, valid_num)


def blend(rgb, weight, bg, thr=-1.0):
    return _Blend.apply(rgb, weight, bg, thr)


def shade(attr, weight, idx, valid_num, bg, thr=-1.0):
    return _Shade.apply(attr, weight, idx, valid_num, bg, thr)


def silhouette(weight):
    return _Silhouette.apply(weight)
