"""Host-side mirror of VoGE/Renderer.py: GaussianRenderer (:87-150), GaussianRenderSettings
(:53-84), Fragments (:13-50), interpolate_attr (:153), get_silhouette (:157-159),
to_colored_background (:162-171), to_white_background (:174-176) -- same names, same argument
meaning.  Every stage behind these calls is a HIP kernel (voge_amd.ops); a renderer on CPU
tensors raises instead of falling back.
"""
import math
import os
from typing import Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .Aggregation import aggregation, expend_sigma, merge_final
from . import RayTracing
from .RayTracing import _view_axis
from .cameras import camera_tensors, pixel_rays

# Fold `verts - origin` and `2 * sigmas` into the trace kernels when the inputs allow it (see forward()).
# VOGE_FUSED_PREAMBLE=0 (or setting this flag) keeps the reference's elementwise torch ops: same results.
FUSED_PREAMBLE = os.environ.get("VOGE_FUSED_PREAMBLE", "1") != "0"


class Fragments(object):
    """vert_weight [.., K] f32, vert_index [.., K] i32, valid_num [..] i64, vert_hit_length [.., K] f32.

    Same fields and methods as VoGE/Renderer.py:13-50.  Fragments made by this renderer from scalar sigmas may arrive
    with their composite DEFERRED (`_lazy`): vert_index and vert_hit_length are there, vert_weight / valid_num are
    computed the first time anybody reads them -- or never, when to_colored_background gets the fragments first and
    produces weights and image in one pass (ops._CompositeShade).  Reading the attributes is all it takes; nothing about
    the values depends on when that happens."""

    _fields = ("vert_weight", "vert_index", "valid_num", "vert_hit_length")

    def __init__(self, vert_weight, vert_index, valid_num, vert_hit_length, _lazy=None):
        self._vert_weight = vert_weight
        self.vert_index = vert_index
        self._valid_num = valid_num
        self._hit_length = vert_hit_length
        self._lazy = _lazy if vert_weight is None else None
        # (the camera-input trace hands out vert_hit_length without a grad_fn; the differentiable alias is made when it is read)
        self._hl_src = _lazy if (_lazy is not None and _lazy.frame and _lazy.sel_len is vert_hit_length) else None
        self._wsum = None      # (per-pixel weight sum left by a one-pass composite + merge: see get_silhouette)

    @property
    def vert_hit_length(self):
        if self._hl_src is not None:
            self._hit_length, self._hl_src = self._hl_src.hit_length(), None
        return self._hit_length

    @vert_hit_length.setter
    def vert_hit_length(self, value):
        self._hit_length, self._hl_src = value, None

    def _composite(self):
        lz, self._lazy = self._lazy, None
        self._vert_weight, self._valid_num = ops.composite_lean(lz)

    def _set_composite(self, weight, valid_num):
        self._lazy = None
        self._vert_weight, self._valid_num = weight, valid_num

    @property
    def vert_weight(self):
        if self._lazy is not None:
            self._composite()
        return self._vert_weight

    @vert_weight.setter
    def vert_weight(self, value):
        if self._lazy is not None and self._valid_num is None:      # (the reference's Fragments always has valid_num)
            self._valid_num = self._lazy.cnt.to(torch.int64)
        self._lazy = None
        self._wsum = None
        self._vert_weight = value

    @property
    def valid_num(self):
        if self._valid_num is None and self._lazy is not None:
            self._valid_num = self._lazy.cnt.to(torch.int64)      # (= the trace's hit count; the composite writes the same)
        return self._valid_num

    @valid_num.setter
    def valid_num(self, value):
        self._valid_num = value

    def _map(self, fn):
        # (views of the same memory keep the trace's bookkeeping -- ops.carry_tags -- so frag.copy(), frag.squeeze()
        #  and frag.unsqueeze() stay on the fast paths, as RenderBunny.py:45's to_white_background(frag.copy(), ...))
        return Fragments(**{k: ops.carry_tags(getattr(self, k), fn(getattr(self, k))) for k in self._fields})

    def __getitem__(self, item):
        assert self.vert_index.dim() == 4, 'Index access is only available when batched.'
        return self._map(lambda t: t[item])

    def __len__(self):
        return self.vert_index.shape[0]

    @property
    def shape(self):
        return tuple(getattr(self, k).shape for k in self._fields)

    def squeeze(self):
        assert self.vert_index.shape[0] == 1
        return self[0]

    def unsqueeze(self):
        assert self.vert_index.dim() == 3
        return self._map(lambda t: t.unsqueeze(0))

    def to_dict(self):
        return {k: getattr(self, k) for k in self._fields}

    def copy(self):
        if self._lazy is not None:      # nothing to copy yet: the same deferred composite (tensors are shared either way,
            return Fragments(None, self.vert_index, self._valid_num, self._hit_length, _lazy=self._lazy)   # .contiguous() is a no-op)
        return self._map(lambda t: t.contiguous())


class GaussianRenderSettings:
    __slots__ = ['image_size', 'max_assign', 'thr_activation', 'absorptivity', 'inverse_sigma', 'principal',
                 'max_point_per_bin']

    def __init__(self, image_size: Union[int, Tuple[int, int]] = 256, max_assign: int = 20,
                 thr_activation: float = 0.01, absorptivity: float = 1, inverse_sigma: bool = False,
                 principal: Union[None, Tuple[int, int], Tuple[float, float]] = None,
                 max_point_per_bin: Union[None, int] = None, **kwargs):
        # unknown keywords (batch_size=, principal_point=, ...) are accepted and ignored, as in
        # Renderer.py:70
        self.image_size = (image_size, image_size) if isinstance(image_size, int) else image_size
        self.max_assign = max_assign
        self.thr_activation = thr_activation
        self.absorptivity = absorptivity
        self.inverse_sigma = inverse_sigma
        self.principal = principal
        self.max_point_per_bin = max_point_per_bin

    def __getitem__(self, item):
        return getattr(self, item)


class GaussianRenderer(nn.Module):
    to_set_args = ['R', 'T', 'focal', 'principal']

    def __init__(self, cameras, render_settings: Union[dict, GaussianRenderSettings]):
        super().__init__()
        self.cameras = cameras
        self.render_settings = render_settings
        self.device = cameras.device
        object.__setattr__(self, "_frame_memo", None)      # (see _frame_camera)

    def to(self, device):
        # cameras are not nn.Modules: move them by hand (Renderer.py:96-100)
        self.cameras = self.cameras.to(device)
        self.device = device
        return self

    def _frame_camera(self, cams, image_size, rows):
        """cameras.camera_tensors with the per-frame constants remembered: in a loop only R and T move, the intrinsics, the
        image size and the band are the same objects every frame -- their preparation (two expands, a cache look-up by
        weak reference, the band arithmetic) was 9 us of every frame's host time."""
        R, T = cams.R, cams.T
        f, p = cams.focal_length, cams.principal_point
        memo = self._frame_memo
        if (memo is not None and memo[0] is f and memo[1] is p and memo[2] is image_size and memo[3] is rows and torch.is_tensor(R)
                and torch.is_tensor(T) and R.dim() == 3 and T.dim() == 2 and R.shape[0] == memo[4] == T.shape[0] and R.is_cuda
                and R.dtype is torch.float32 is T.dtype and T.device == R.device and not (R.requires_grad or T.requires_grad)
                and (not torch.is_tensor(f) or f._version == memo[5]) and (not torch.is_tensor(p) or p._version == memo[6])
                and not torch.cuda.is_current_stream_capturing()):
            return (R, T) + memo[7]
        cam = camera_tensors(cams, image_size, rows)
        object.__setattr__(self, "_frame_memo", None)
        if (cam is not None and cam[0] is R and cam[1] is T and not torch.cuda.is_current_stream_capturing()
                and (rows is None or isinstance(rows, tuple) or hasattr(rows, "stripe_h"))):
            # (remembered by identity: the objects are held here, so an id cannot be recycled while the memo lives)
            object.__setattr__(self, "_frame_memo", (f, p, image_size, rows, R.shape[0], f._version if torch.is_tensor(f) else 0,
                                                     p._version if torch.is_tensor(p) else 0, cam[2:]))
        return cam

    def forward(self, gmeshes, **kwargs):
        """gmeshes() -> (verts [N,3] | [B,N,3], sigmas [N] | [N,3] | [N,3,3], radians);
        R=, T= (and the inert focal=, principal=) keywords are stored on the camera object
        (Renderer.py:104-109).  `rows=(r0, r1)` (extension) renders only that pixel-row band."""
        cams = self.cameras
        assert not cams.in_ndc(), 'Got NDC camera. Cameras.in_ndc must be set to false.'
        for name in self.to_set_args:
            if name in kwargs:
                v = kwargs[name]
                setattr(cams, name, v.to(self.device) if isinstance(v, torch.Tensor) else v)
        st = self.render_settings
        image_size = st['image_size']

        verts, sigmas, _radians = gmeshes()
        shared_verts = verts.dim() == 2
        verts2d = verts                      # (the [N,3] parameter itself: indexing it back out of verts[None] would put a
        if shared_verts:                     #  select + zero-fill + copy into every backward)
            verts = verts[None]

        thr_act = -math.log(st['thr_activation'] + 1 / 1e10)                     # RayTracing.py:76,85
        K, occ = st['max_assign'], st['absorptivity']
        behind = st['max_point_per_bin'] != -1      # the coarse stage's "skip z < 0" candidate rule (rasterize_coarse.cu:35)
        if sigmas.dim() == 1 and shared_verts and FUSED_PREAMBLE and not (behind and RayTracing.REFERENCE_CANDIDATES):
            # Round 6, the frame path: one (verts [N,3], sigmas [N]) set, fixed cameras -- the trace takes the CAMERA itself
            # (ops._FrameTrace: no ray-generation launch, no ray node; rays, cones, camera centre and view axis are made inside
            # binA / binB / the sweep with the ray kernel's own operations) and stops behind the sweep like trace_lean
            cam = self._frame_camera(cams, image_size, kwargs.get('rows'))
            if cam is not None and ops.frame_eligible(verts2d, sigmas, *cam[:4], K, cam[4][1] * cam[5]):
                index, hit_len, lz = ops.frame_trace(verts2d, sigmas, *cam[:4], cam[4], cam[5], behind, thr_act, K,
                                                     2 if st['inverse_sigma'] else 1, occ)
                return Fragments(None, index, None, hit_len, _lazy=lz)
        if (sigmas.dim() >= 2 and sigmas.shape[-1] == 3 and not st['inverse_sigma'] and FUSED_PREAMBLE and verts.is_cuda
                and not (behind and RayTracing.REFERENCE_CANDIDATES) and os.environ.get("VOGE_LAZY_GENERAL", "1") != "0"):
            # (N,3) / (N,3,3) sigmas on the frame path: the camera AND the user's own arrays go into the trace, whose record pass
            # does the centring and 2 * expend_sigma of Renderer.py:130-137 (no ray launch, no preamble launch, no node here)
            cam = self._frame_camera(cams, image_size, kwargs.get('rows'))
            if cam is not None and ops.frame_eligible(verts2d, sigmas, *cam[:4], K, cam[4][1] * cam[5]):
                index, hit_len, lz = ops.frame_trace_gen(verts2d, sigmas, *cam[:4], cam[4], cam[5], behind, thr_act, K, occ)
                return Fragments(None, index, None, hit_len, _lazy=lz)
        rays, origin = pixel_rays(cams, image_size, rows=kwargs.get('rows'))     # [B,h,W,3], [B,3]
        # ray_tracing (RayTracing.py:12-30) + aggregation (Aggregation.py:82-107) as ONE call: the trace's sweep
        # composites the fragments in its epilogue (voge_fragments_fwd*).  The stand-alone ray_tracing* / aggregation
        # functions remain the public API and produce the same values.
        if behind and RayTracing.REFERENCE_CANDIDATES:
            # the reference's own coarse candidate lists (lossy on purpose): explicit lists, unfused calls
            sig3 = expend_sigma(sigmas)
            if sig3.dim() == 3:
                sig3 = sig3.unsqueeze(0).expand(verts.shape[0] if not shared_verts else origin.shape[0], -1, -1, -1)
            isigma = 2 * torch.inverse(sig3) if st['inverse_sigma'] else 2 * sig3
            centred = verts - origin[:, None]
            sel_idx, sel_len, sel_act, sel_dsd = RayTracing.ray_tracing(
                cams, centred, isigma.contiguous(), rays, image_size, thr=st['thr_activation'], n_assign=K,
                max_points_per_bin=st['max_point_per_bin'])
            weight, index, valid_num, hit_len = aggregation(sel_idx, sel_act, sel_len, sel_dsd, occ)
            return Fragments(vert_weight=weight, vert_index=index, valid_num=valid_num, vert_hit_length=hit_len)
        if sigmas.dim() == 1 and shared_verts and not origin.requires_grad and FUSED_PREAMBLE:
            # One (verts [N,3], sigmas [N]) set seen by every view, fixed cameras: the centring of
            # Renderer.py:130 and the 2*sigma / 2/sigma of :133-137 happen inside the trace's per-Gaussian
            # pass (and their chain rule inside its backward's) -- same values, no elementwise launches.
            cam_fwd = _view_axis(cams, origin[:, None]) if behind else None
            smode = 2 if st['inverse_sigma'] else 1
            if ops.lazy_eligible(2, verts2d, sigmas, origin, rays, K):
                # stop behind the sweep: the composite runs when the weights are first read -- or inside
                # to_colored_background's own pass (Fragments._lazy)
                index, hit_len, lz = ops.trace_lean(2, verts2d, sigmas, origin, rays, cam_fwd, thr_act, K, smode, occ)
                return Fragments(None, index, None, hit_len, _lazy=lz)
            weight, index, valid_num, hit_len = ops.fragments(2, verts2d, sigmas, origin, rays, cam_fwd, thr_act, K, smode, occ)
            return Fragments(vert_weight=weight, vert_index=index, valid_num=valid_num, vert_hit_length=hit_len)
        if (sigmas.dim() >= 2 and not st['inverse_sigma'] and not origin.requires_grad and FUSED_PREAMBLE
                and verts.is_cuda and sigmas.shape[-1] == 3):
            # (N,3) / (N,3,3) sigmas: the centring and 2 * expend_sigma of Renderer.py:130-137 as ONE launch each way
            cam_fwd = _view_axis(cams, origin[:, None]) if behind else None
            mus0, isg0 = ops.general_preamble(verts2d if shared_verts else verts, sigmas, origin)
            if ops.lazy_eligible(0, mus0, isg0, None, rays, K):
                index, hit_len, lz = ops.trace_lean(0, mus0, isg0, None, rays, cam_fwd, thr_act, K, 0, occ)
                return Fragments(None, index, None, hit_len, _lazy=lz)
            weight, index, valid_num, hit_len = ops.fragments(0, mus0, isg0, None, rays, cam_fwd, thr_act, K, 0, occ)
            return Fragments(vert_weight=weight, vert_index=index, valid_num=valid_num, vert_hit_length=hit_len)
        centred = verts - origin[:, None]                                         # Renderer.py:130
        cam_fwd = _view_axis(cams, centred) if behind else None
        B = centred.shape[0]
        if sigmas.dim() == 1:
            # (N,) sigmas are isotropic: expend_sigma would give sigma * I (Aggregation.py:155-157) and
            # Renderer.py:133 doubles (or inverts and doubles) it.  Keep the scalar: the trace has an
            # isotropic form whose backward produces d/d(scalar) directly.
            a = 2.0 / sigmas if st['inverse_sigma'] else 2.0 * sigmas
            a = a.unsqueeze(0).expand(B, -1)
            mus1, a1 = centred.reshape(-1, 3), a.reshape(-1)
            if ops.lazy_eligible(1, mus1, a1, None, rays, K):
                index, hit_len, lz = ops.trace_lean(1, mus1, a1, None, rays, cam_fwd, thr_act, K, 0, occ)
                return Fragments(None, index, None, hit_len, _lazy=lz)
            weight, index, valid_num, hit_len = ops.fragments(1, mus1, a1, None, rays, cam_fwd, thr_act, K, 0, occ)
        else:
            sigmas = expend_sigma(sigmas)
            if sigmas.dim() == 3:
                sigmas = sigmas.unsqueeze(0).expand(B, -1, -1, -1)
            isigma = 2 * torch.inverse(sigmas) if st['inverse_sigma'] else 2 * sigmas
            mus0, isg0 = centred.reshape(-1, 3), isigma.reshape(-1, 3, 3)
            if ops.lazy_eligible(0, mus0, isg0, None, rays, K):
                # (full 3x3 forms defer their composite too: the sweep keeps the packed (mu, A) records instead of act / dsd)
                index, hit_len, lz = ops.trace_lean(0, mus0, isg0, None, rays, cam_fwd, thr_act, K, 0, occ)
                return Fragments(None, index, None, hit_len, _lazy=lz)
            weight, index, valid_num, hit_len = ops.fragments(0, mus0, isg0, None, rays, cam_fwd, thr_act, K, 0, occ)
        # merge_final later rewrites -1 -> 0 inside the fragments' index tensor.  The reference clones
        # it here (Renderer.py:145) because its backward finds empty slots by idx == -1; this trace
        # backward uses the per-pixel hit count instead, so no copy is needed.
        return Fragments(vert_weight=weight, vert_index=index, valid_num=valid_num, vert_hit_length=hit_len)


def interpolate_attr(fragments: Fragments, vert_attr: torch.Tensor):
    lz = getattr(fragments, "_lazy", None)
    if lz is not None:
        # fragments whose composite is still pending: weights, merged attributes and the per-pixel weight sum in one
        # pass (ops._CompositeMerge); a get_silhouette on the same fragments then costs nothing but a clamp
        assert vert_attr.dim() == 2
        out = ops.composite_merge(lz, vert_attr)
        if out is not None:
            fragments._set_composite(out[2], out[3])
            fragments._wsum = (out[1], out[2], out[2]._version, out[4])
            return out[0]
    return merge_final(vert_attr=vert_attr, weight=fragments.vert_weight, valid_num=fragments.valid_num,
                       vert_assign=fragments.vert_index)


def get_silhouette(fragments: Fragments):
    ws = getattr(fragments, "_wsum", None)
    if ws is not None and ws[1] is fragments._vert_weight and ws[1]._version == ws[2]:
        if ws[3] is not None:      # (frame path: the composite wrote min(sum_k w_k, 1) itself -- an output of the same node as the merge)
            return ws[3]
        # the weight sum the one-pass composite + merge left behind: min(sum_k w_k, 1) exactly as Renderer.py:157-159
        # (torch.minimum splits the gradient at a tie like torch.min(a, b))
        return torch.minimum(ws[0], torch.ones_like(ws[0]))
    return ops.silhouette(fragments.vert_weight)


_BG_CACHE = {}


def _background_tensor(color, device):
    """Device tensor of a constant background colour, uploaded once per (colour, device): a
    host-to-device copy per frame would also make the frame impossible to capture in a HIP graph."""
    key = (color if type(color) is tuple else tuple(color), device)      # ((1, 1, 1) and (1.0, 1.0, 1.0) hash and compare equal)
    t = _BG_CACHE.get(key)
    if t is None:
        t = _BG_CACHE[key] = torch.tensor(tuple(float(c) for c in color), dtype=torch.float32, device=device)
    return t


def to_colored_background(fragments: Fragments, colors: torch.Tensor,
                          background_color: Union[torch.Tensor, tuple, list] = (1, 1, 1), thr: float = -1):
    if not torch.is_tensor(background_color):
        background_color = _background_tensor(background_color, colors.device)
    elif background_color.device != colors.device:
        background_color = background_color.to(colors.device)
    lz = getattr(fragments, "_lazy", None)
    if lz is not None:
        # fragments whose composite is still pending: weights AND image in one pass (ops._CompositeShade)
        out = ops.composite_shade(lz, colors, background_color, thr)
        if out is not None:
            fragments._set_composite(out[1], out[2])
            return out[0]
    if colors.dim() == 2 and colors.shape[1] <= 4:
        # merge + silhouette + blend fused in one kernel; on fragments of this renderer the backward of the whole
        # pipeline (this blend, the composite, the trace) is one kernel as well (ops._ShadeThrough)
        img = ops.shade_through(colors, fragments.vert_weight, fragments.vert_index, fragments.valid_num, background_color, thr)
        if img is not None:
            return img
        return ops.shade(colors, fragments.vert_weight, fragments.vert_index, fragments.valid_num, background_color, thr)
    rgb = interpolate_attr(fragments, colors)
    return ops.blend(rgb, fragments.vert_weight, background_color, thr)


def to_white_background(fragments: Fragments, colors: torch.Tensor, thr: float = -1):
    return to_colored_background(fragments=fragments, colors=colors, background_color=(1, 1, 1), thr=thr)
