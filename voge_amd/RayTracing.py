"""Host-side mirror of VoGE/RayTracing.py for the MI355X build (same names, same argument
meaning, same error behaviour for the on-path entry points).

ray_tracing          <- VoGE/RayTracing.py:12-30
ray_tracing_fine     <- VoGE/RayTracing.py:76-95
_RayTraceVoGE        <- VoGE/RayTracing.py:154-206  (implemented in voge_amd.ops)

Difference by design (DESIGN.md §Candidate sets): the reference either materialises an
all-Gaussians list per bin (max_points_per_bin == -1) or runs a lossy coarse rasteriser
(bin overflow drops candidates, rasterize_coarse.cu:154-170).  Here every setting runs the
same exact sweep -- culling happens inside the kernel and is conservative -- and
max_points_per_bin != -1 only adds the coarse stage's "skip Gaussians behind the camera"
rule (rasterize_coarse.cu:35).  On BASELINE configs 1 and 2 that yields the reference's fragments
exactly (tests/test_gpu_next_rows.py).  REFERENCE_CANDIDATES = True switches to the reference's
own candidate lists (rasterize_coarse below: its bounding boxes, bins and chunk-overflow drop),
for scenes where its lossy lists are the behaviour wanted.
"""
import math
from typing import Optional

import numpy as np
import torch

from .ops import _FindNearestK, _RayTraceVoGE, _RayTraceVoGEIso, _RayTraceVoGEIsoView, _RayTraceVoGERay

inf = 1e8

# False: conservative in-kernel culling (+ the "skip z < 0" rule); True: the reference's coarse candidate lists
REFERENCE_CANDIDATES = False


def default_bin_size(image_size):
    """RayTracing.py:14-16."""
    return max(int(2 ** np.ceil(np.log2(max(image_size)) - 5)), 10)


def ray_tracing(transforms, points, isigmas, rays, image_size, thr: float, n_assign: int,
                bin_size: Optional[int] = None, max_points_per_bin: Optional[int] = None, **kwargs):
    """points [B,N,3] camera-centred, isigmas [B,N,3,3], rays [B,H,W,3] -> sel_idx, sel_len,
    sel_act, sel_dsd, each [B,H,W,n_assign]."""
    if bin_size is None:
        bin_size = default_bin_size(image_size)
    if max_points_per_bin is None and REFERENCE_CANDIDATES:
        max_points_per_bin = min(int(max(n_assign * 10, (points.shape[1]) / 10)), points.shape[1])      # RayTracing.py:19
    if max_points_per_bin == -1:
        candidates = None
    elif REFERENCE_CANDIDATES:
        candidates = rasterize_coarse(transforms, points, isigmas, image_size, thr, bin_size, max_points_per_bin, **kwargs)
    else:
        # coarse-stage candidate rule kept: view-space z >= 0 (rasterize_coarse.cu:35).  The view
        # axis in the rays' (world-aligned) frame is the third column of R (X_view = X_world @ R + T).
        candidates = _view_axis(transforms, points)
    return ray_tracing_fine(points.reshape(-1, 3), isigmas.reshape(-1, 3, 3), rays, candidates, thr, bin_size,
                            n_assign)


def ray_tracing_iso(transforms, points, a, rays, image_size, thr: float, n_assign: int,
                    max_points_per_bin: Optional[int] = None, inf=1e10, **kwargs):
    """ray_tracing for isotropic Gaussians kept in scalar form: points [B,N,3] camera-centred,
    a [N] (shared by the batch) or [B,N] with A = a I.  Same outputs as ray_tracing(points,
    a[..., None, None] * eye(3), ...); the gradient reaches `a` directly."""
    B, N = points.shape[0], points.shape[1]
    if a.dim() == 1:
        a = a.unsqueeze(0).expand(B, -1)
    candidates = None if max_points_per_bin == -1 else _view_axis(transforms, points)
    thr_act = -math.log(thr + 1 / inf)
    return _RayTraceVoGEIso.apply(points.reshape(-1, 3), a.reshape(-1), rays, candidates, thr_act, n_assign)


def ray_tracing_iso_view(transforms, verts, sigmas, origin, rays, image_size, thr: float, n_assign: int,
                         max_points_per_bin: Optional[int] = None, inverse_sigma: bool = False, inf=1e10, **kwargs):
    """ray_tracing_iso with the renderer's preamble folded into the kernels: verts [N,3] and sigmas [N]
    as the user holds them (shared by all B views), origin [B,3] the camera centres; equivalent to
    ray_tracing_iso(verts[None] - origin[:, None], 2 * sigmas or 2 / sigmas, ...) (Renderer.py:130-137),
    bit-identical outputs, without the three elementwise launches and their backward.  origin gets no
    gradient here."""
    candidates = None if max_points_per_bin == -1 else _view_axis(transforms, origin[:, None])
    thr_act = -math.log(thr + 1 / inf)
    return _RayTraceVoGEIsoView.apply(verts, sigmas, origin, rays, candidates, thr_act, n_assign,
                                      2 if inverse_sigma else 1)


def convert_to_box(isigmas, thr, z, matrix):
    """RayTracing.py:33-39: per-axis half extents sqrt(colsum(-ln(thr) F A2^-1 F)) * z, A2 the upper-left 2x2 block of
    the view-space precision (the block's inverse, not the marginal: exact for isotropic A), F = matrix[:, :2, :2]."""
    get = -np.log(thr) * matrix[:, None, :2, :2] @ torch.inverse(isigmas[:, :, :2, :2]) @ matrix[:, None, :2, :2]
    return (torch.ones((*isigmas.shape[0:2], 1, 2), device=isigmas.device) @ get).pow(.5).squeeze(2) * z.unsqueeze(-1)


def rasterize_coarse(cameras, points, isigmas, image_size, thr, bin_size, max_points_per_bin, cloud_to_point=None,
                     num_points_per_cloud=None):
    """The reference's coarse stage (RayTracing.py:42-73): camera-centred points [B,N,3] and A [B,N,3,3] -> candidate
    lists [B,BH,BW,M] int32.  The PyTorch3D transforms it composes (get_full_projection_transform, then
    get_ndc_camera_transform, negated) are written out for the screen-space pinhole camera: with view axes +X left /
    +Y up the centre projects to column px - fx X/Z, row py - fy Y/Z, and the negated NDC is (col - W/2) 2/s,
    (row - H/2) 2/s, s = min(H, W); F = diag(-2 fx/s, -2 fy/s)."""
    from . import ops
    from .cameras import _as_b2
    B, N = points.shape[0], points.shape[1]
    dev = points.device
    R = torch.as_tensor(cameras.R, dtype=torch.float32, device=dev).reshape(-1, 3, 3).expand(B, 3, 3)
    T = torch.as_tensor(cameras.T, dtype=torch.float32, device=dev).reshape(-1, 3).expand(B, 3)
    f, pp = _as_b2(cameras.focal_length, B, dev), _as_b2(cameras.principal_point, B, dev)
    H, W = int(image_size[0]), int(image_size[1])
    s = float(min(H, W))
    C = -torch.matmul(torch.inverse(R.transpose(1, 2)), T[:, :, None])                  # :45
    world = points + C.view(-1, 1, 3)                                                    # :46
    view = world @ R + T[:, None]                                                         # X_view = X_world R + T
    z = view[..., 2]
    cols = pp[:, None, 0] - f[:, None, 0] * view[..., 0] / z
    rows = pp[:, None, 1] - f[:, None, 1] * view[..., 1] / z
    points_ndc = torch.stack([(cols - W / 2.0) * (2.0 / s), (rows - H / 2.0) * (2.0 / s), z], dim=-1)      # :50, :57
    rot = R[:, None].expand(-1, N, -1, -1)
    isig_view = rot.transpose(2, 3) @ isigmas @ rot                                      # :52-53
    F = torch.zeros((B, 4, 4), dtype=torch.float32, device=dev)
    F[:, 0, 0], F[:, 1, 1] = -2.0 * f[:, 0] / s, -2.0 * f[:, 1] / s
    boxes = convert_to_box(isig_view, thr, 1.0 / z, F)                                   # :55 (z argument = -ndc z = 1/Z)
    if cloud_to_point is None:
        cloud_to_point = torch.arange(B, dtype=torch.long, device=dev) * N
    if num_points_per_cloud is None:
        num_points_per_cloud = torch.ones(B, dtype=torch.long, device=dev) * N
    return ops.rasterize_points_coarse(points_ndc.reshape(-1, 3), cloud_to_point, num_points_per_cloud, (H, W),
                                       boxes.reshape(-1, 2), bin_size, max_points_per_bin)


def _view_axis(cameras, points):
    R = getattr(cameras, "R", None)
    if R is None:
        return None
    R = torch.as_tensor(R, dtype=torch.float32, device=points.device).reshape(-1, 3, 3)
    B = points.shape[0]
    if R.shape[0] != B:
        R = R.expand(B, -1, -1)
    return R[:, :, 2].detach().contiguous()


def ray_tracing_fine(mus, isigmas, rays, bin_points, thr, bin_size, n_assign, inf=1e10):
    assert isigmas.dim() == 3
    assert mus.dim() == 2
    assert rays.dim() == 4
    assert bin_points is None or bin_points.dim() in (2, 4)
    assert mus.shape[0] == isigmas.shape[0] and mus.shape[1] == 3 and isigmas.shape[1] == 3 and isigmas.shape[2] == 3
    thr_act = -math.log(thr + 1 / inf)
    return _RayTraceVoGE.apply(mus, isigmas, rays, bin_points, thr_act, bin_size, n_assign)


def ray_trace_voge_ray(mus, sigmas, rays):
    """Dense (N rays x M Gaussians) trace, RayTracing.py:97-108: sigmas may be a float, (M,) or (M,3,3)."""
    if isinstance(sigmas, (float, int)):
        sigmas = torch.eye(3, device=mus.device)[None].expand(mus.shape[0], -1, -1) * sigmas
    if sigmas.dim() == 1:
        sigmas = sigmas.view(-1, 1, 1) * torch.eye(3, device=sigmas.device)[None]
    assert mus.is_cuda and sigmas.is_cuda and rays.is_cuda
    assert mus.dim() == 2 and mus.shape[1] == 3
    assert rays.dim() == 2 and rays.shape[1] == 3
    assert sigmas.dim() == 3 and sigmas.shape[1] == 3 and sigmas.shape[2] == 3
    return _RayTraceVoGERay.apply(mus, sigmas, rays)


def find_nearest_k(hit_len_in, hit_act_in, hit_dsd_in, K, thr):
    """RayTracing.py:111-115 (note: this entry point uses the module-level inf = 1e8)."""
    assert hit_len_in.is_cuda and hit_act_in.is_cuda and hit_dsd_in.is_cuda
    thr_act = -math.log(thr + 1 / inf)
    return _FindNearestK.apply(hit_len_in, hit_act_in, hit_dsd_in, thr_act, K)


def find_farest_k(hit_len_in, hit_act_in, hit_dsd_in, K, thr):
    """RayTracing.py:118-123: nearest-K on the negated lengths."""
    assert hit_len_in.is_cuda and hit_act_in.is_cuda and hit_dsd_in.is_cuda
    thr_act = -math.log(thr + 1 / inf)
    point_idx, hit_len, hit_act, hit_dsd = _FindNearestK.apply(-hit_len_in, hit_act_in, hit_dsd_in, thr_act, K)
    return point_idx, -hit_len, hit_act, hit_dsd
