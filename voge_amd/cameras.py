"""Camera shim + ray generation for the VoGE hot path.

The reference takes PyTorch3D camera objects and calls
`NDCMultinomialRaysampler(..., unit_directions=True)(cameras)` (VoGE/Renderer.py:124-128).
PyTorch3D is an un-vendored dependency that is absent from this image, so this module
provides (a) `pixel_rays`, which produces the same bundle from ANY object exposing
`R [B,3,3]`, `T [B,3]`, `focal_length`, `principal_point` (real PyTorch3D screen-space
`PerspectiveCameras` included, by duck typing), and (b) a minimal `PerspectiveCameras` /
`look_at_view_transform` pair with PyTorch3D's conventions so the demo scripts' camera code
has something to import where PyTorch3D is not installed.

Conventions (PyTorch3D, screen space, in_ndc=False): row vectors, X_view = X_world @ R + T;
view axes +X left, +Y up, +Z forward; pixel (row i, col j) centre <-> view direction
[(px - j - 0.5)/fx, (py - i - 0.5)/fy, 1]; rays leave the camera centre C = -T @ R^-1.
Ray generation is a HIP kernel with an analytic backward (gradients reach R, T, focal_length,
principal_point).
"""
import math

import numpy as np
import torch


def _as_b2(v, B, device, dtype=torch.float32):
    """focal_length / principal_point -> [B,2].  PyTorch3D reads a 1-D tensor of length B as one value per
    camera (fx = fy); only a [.,2] shape (or a 1-D pair for a single camera) means (x, y)."""
    v = torch.as_tensor(v, dtype=dtype, device=device)
    if v.dim() == 0:
        v = v.reshape(1, 1)
    elif v.dim() == 1:
        per_camera = (v.shape[0] == B and B != 1) if B else False      # [B] scalars, including B == 2
        v = v.reshape(-1, 1) if (per_camera or v.shape[0] != 2) else v.reshape(1, 2)
    if v.shape[-1] == 1:
        v = v.expand(-1, 2)
    if v.shape[0] != B:
        v = v.expand(B, 2)
    return v


_INTR = {}      # (one entry per device: the last (focal_length, principal_point) pair prepared and what it became)


def _intrinsics(focal, pp, B, device):
    """focal_length / principal_point as contiguous fp32 [B,2] tensors.  The same two objects come in on every frame of a loop
    (only R / T move), so the last pair is remembered per device -- by identity and version of the tensors, held weakly --
    instead of being re-expanded and re-copied by two tiny kernels per frame."""
    import weakref
    key = str(device)
    # (while a HIP graph is being captured the cache is neither read nor written, like ops._workspace: a pair prepared inside
    # a capture lives in the graph's private pool and is only filled when the graph replays -- an eager call that hit such an
    # entry would read unwritten memory)
    capturing = torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()
    hit = None if capturing else _INTR.get(key)
    if (hit is not None and torch.is_tensor(focal) and torch.is_tensor(pp) and hit[0]() is focal and hit[1]() is pp
            and hit[2] == (focal._version, pp._version, B) and not focal.requires_grad and not pp.requires_grad):
        return hit[3], hit[4]
    f_c, p_c = _as_b2(focal, B, device), _as_b2(pp, B, device)
    if torch.is_tensor(focal) and torch.is_tensor(pp) and not focal.requires_grad and not pp.requires_grad:
        f_c, p_c = f_c.contiguous(), p_c.contiguous()
        # (an input that already is a contiguous [B,2] fp32 tensor comes back as itself: caching it would hold a strong
        # reference the weak ones never release -- and there is nothing to save)
        if not capturing and f_c is not focal and p_c is not pp:
            _INTR[key] = (weakref.ref(focal), weakref.ref(pp), (focal._version, pp._version, B), f_c, p_c)
    return f_c, p_c


class PerspectiveCameras:
    """Screen-space pinhole cameras: the subset of pytorch3d.renderer.PerspectiveCameras the
    renderer and the demos touch (attributes R, T, focal_length, principal_point, image_size,
    device; in_ndc(); to(); get_camera_center())."""

    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), R=None, T=None, device="cpu",
                 in_ndc=False, image_size=None):
        self.device = torch.device(device)
        self.R = torch.eye(3)[None] if R is None else torch.as_tensor(R, dtype=torch.float32).reshape(-1, 3, 3)
        self.T = torch.zeros(1, 3) if T is None else torch.as_tensor(T, dtype=torch.float32).reshape(-1, 3)
        B = max(self.R.shape[0], self.T.shape[0])
        self.focal_length = _as_b2(focal_length, 1, "cpu") if not torch.is_tensor(focal_length) else focal_length
        self.principal_point = _as_b2(principal_point, 1, "cpu") if not torch.is_tensor(principal_point) else principal_point
        self.image_size = None if image_size is None else torch.as_tensor(image_size).reshape(-1, 2)
        self._in_ndc = bool(in_ndc)
        self._B = B
        self.to(self.device)

    def in_ndc(self):
        return self._in_ndc

    def __len__(self):
        return max(self.R.shape[0], self.T.shape[0])

    def to(self, device):
        self.device = torch.device(device)
        for name in ("R", "T", "focal_length", "principal_point", "image_size"):
            v = getattr(self, name)
            if torch.is_tensor(v):
                setattr(self, name, v.to(self.device))
        return self

    def get_camera_center(self):
        R = self.R
        return -torch.einsum("bj,bjk->bk", self.T.expand(R.shape[0], 3), torch.linalg.inv(R))


def pixel_rays(cameras, image_size, rows=None):
    """Unit world-space ray directions [B,h,W,3] for pixel rows `rows=(r0,r1)` (default: all H
    rows) and the camera centres [B,3]: what Renderer.py:124-128 reads from the ray bundle.
    `rows` may also be a distributed.Stripes (a rank's interleaved stripes of the frame, stacked into one image).
    One HIP kernel (voge_rays_fwd); differentiable w.r.t. R, T, focal_length, principal_point."""
    from . import ops
    H, W = int(image_size[0]), int(image_size[1])
    R = cameras.R
    device = R.device
    R = R.to(torch.float32).reshape(-1, 3, 3)
    T = torch.as_tensor(cameras.T, dtype=torch.float32, device=device).reshape(-1, 3)
    B = max(R.shape[0], T.shape[0])
    R = R.expand(B, 3, 3)
    T = T.expand(B, 3)
    f, pp = _intrinsics(cameras.focal_length, cameras.principal_point, B, device)
    if hasattr(rows, "stripe_h"):      # distributed.Stripes
        return ops.pixel_rays(R, T, f, pp, rows.row0, rows.h, W, rows.stripe_h, rows.pitch)
    r0, r1 = (0, H) if rows is None else (int(rows[0]), int(rows[1]))
    return ops.pixel_rays(R, T, f, pp, r0, r1 - r0, W)


def camera_tensors(cameras, image_size, rows=None):
    """What the camera-input trace (ops._FrameTrace) takes: (R [B,3,3], T [B,3], focal [B,2], pp [B,2], band, W) with
    band = (row0, h, stripe_h, pitch) -- pixel_rays' arguments without the ray kernel -- or None when the camera does not
    fit that path (a pose or an intrinsic that wants a gradient, tensors on another device)."""
    R, T = cameras.R, cameras.T
    if not (torch.is_tensor(R) and torch.is_tensor(T)) or R.requires_grad or T.requires_grad or not R.is_cuda:
        return None
    if R.dim() != 3 or T.dim() != 2 or R.dtype != torch.float32 or T.dtype != torch.float32 or T.device != R.device:
        R = R.to(torch.float32).reshape(-1, 3, 3)
        T = torch.as_tensor(T, dtype=torch.float32, device=R.device).reshape(-1, 3)
    B = max(R.shape[0], T.shape[0])
    if R.shape[0] != B or T.shape[0] != B:
        R, T = R.expand(B, 3, 3), T.expand(B, 3)
    f, pp = _intrinsics(cameras.focal_length, cameras.principal_point, B, R.device)
    if f.requires_grad or pp.requires_grad:
        return None
    H, W = int(image_size[0]), int(image_size[1])
    if hasattr(rows, "stripe_h"):      # distributed.Stripes
        band = (int(rows.row0), int(rows.h), int(rows.stripe_h), int(rows.pitch))
    else:
        r0, r1 = (0, H) if rows is None else (int(rows[0]), int(rows[1]))
        band = (r0, r1 - r0, max(r1 - r0, 1), 0)
    return R, T, f, pp, band, W


def camera_position_from_spherical_angles(distance, elevation, azimuth, degrees=True, device="cpu"):
    d, e, a = torch.broadcast_tensors(*(torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1)
                                        for v in (distance, elevation, azimuth)))
    if degrees:
        e, a = e * (math.pi / 180.0), a * (math.pi / 180.0)
    return torch.stack([d * torch.cos(e) * torch.sin(a), d * torch.sin(e), d * torch.cos(e) * torch.cos(a)], dim=-1)


def look_at_rotation(camera_position, at=((0, 0, 0),), up=((0, 1, 0),), device="cpu"):
    C = torch.as_tensor(camera_position, dtype=torch.float32, device=device).reshape(-1, 3)
    at = torch.as_tensor(at, dtype=torch.float32, device=device).reshape(-1, 3).expand(C.shape[0], 3)
    up = torch.as_tensor(up, dtype=torch.float32, device=device).reshape(-1, 3).expand(C.shape[0], 3)
    nrm = lambda v: torch.nn.functional.normalize(v, eps=1e-5, dim=-1)
    z = nrm(at - C)
    x = nrm(torch.cross(up, z, dim=1))
    y = nrm(torch.cross(z, x, dim=1))
    degenerate = torch.isclose(x, torch.zeros_like(x), atol=5e-3).all(dim=1, keepdim=True)
    x = torch.where(degenerate, nrm(torch.cross(y, z, dim=1)), x)
    return torch.stack([x, y, z], dim=-1)  # x, y, z as columns


def _look_at_host(dist, elev, azim, degrees, eye, at, up):
    """look_at_view_transform for plain numbers / lists / arrays, on the host in float32 (the same operations in the same
    order as the tensor path below) -> R [B,3,3], T [B,3] as numpy arrays."""
    f32 = np.float32
    at = np.asarray(at, f32).reshape(-1, 3)
    if eye is not None:
        C = np.asarray(eye, f32).reshape(-1, 3)
    else:
        d, e, a = np.broadcast_arrays(*(np.asarray(v, f32).reshape(-1) for v in (dist, elev, azim)))
        if degrees:
            e, a = e * f32(math.pi / 180.0), a * f32(math.pi / 180.0)
        C = np.stack([d * np.cos(e) * np.sin(a), d * np.sin(e), d * np.cos(e) * np.cos(a)], axis=-1).astype(f32) + at
    C, at = np.broadcast_arrays(C, at)      # (one eye with several targets broadcasts like the tensor path)
    B = C.shape[0]
    up = np.broadcast_to(np.asarray(up, f32).reshape(-1, 3), (B, 3))
    nrm = lambda v: v / np.maximum(np.sqrt((v * v).sum(-1, keepdims=True, dtype=f32)), f32(1e-5))
    z = nrm(at - C)
    x = nrm(np.cross(up, z).astype(f32))
    y = nrm(np.cross(z, x).astype(f32))
    degenerate = (np.abs(x) <= 5e-3).all(axis=1, keepdims=True)
    x = np.where(degenerate, nrm(np.cross(y, z).astype(f32)), x)
    R = np.stack([x, y, z], axis=-1).astype(f32)
    T = -np.einsum("bji,bj->bi", R, C).astype(f32)
    return R, T


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, eye=None, at=((0, 0, 0),),
                           up=((0, 1, 0),), device="cpu"):
    """(R [B,3,3], T [B,3]) with PyTorch3D's spherical convention:
    C = dist*[cos(e)sin(a), sin(e), cos(e)cos(a)] + at,  T = -R^T C.
    Arguments that are all plain numbers / sequences are evaluated on the host and R, T reach the device in ONE copy (as tensors
    on the device every scalar is its own host-to-device copy and every operation a launch: ~ 0.67 ms per call, more than a
    frame -- a training loop that draws a view per iteration spends its time here); tensors take the tensor path, which is
    differentiable."""
    if not any(isinstance(v, torch.Tensor) for v in (dist, elev, azim, eye, at, up)):
        R, T = _look_at_host(dist, elev, azim, degrees, eye, at, up)
        B = R.shape[0]
        # (ONE host-to-device copy; T starts at a 16-byte boundary of the shared buffer, so a vectorised load of either is fine)
        t0 = (B * 9 + 3) // 4 * 4
        host = np.zeros(t0 + B * 3, np.float32)
        host[:B * 9] = R.reshape(-1)
        host[t0:] = T.reshape(-1)
        buf = torch.from_numpy(host).to(device)
        return buf[:B * 9].view(B, 3, 3), buf[t0:].view(B, 3)
    if eye is not None:
        C = torch.as_tensor(eye, dtype=torch.float32, device=device).reshape(-1, 3)
    else:
        C = camera_position_from_spherical_angles(dist, elev, azim, degrees=degrees, device=device)
        C = C + torch.as_tensor(at, dtype=torch.float32, device=device).reshape(-1, 3)
    R = look_at_rotation(C, at=at, up=up, device=device)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T
