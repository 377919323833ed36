"""numpy restatement of the camera / ray conventions on the path -- TEST INFRASTRUCTURE.

The reference delegates this arithmetic to PyTorch3D (un-vendored, unpinned:
``VoGE/Renderer.py:7,124-128``), which is not present in
/root/reference nor in this image, so this row is "parity unpinned": the convention below
is the published PyTorch3D screen-space ``PerspectiveCameras`` +
``NDCMultinomialRaysampler(unit_directions=True)`` behaviour, restated from its
documentation, and is self-consistent with the reference's own ``PixToNonSquareNdc``
(``VoGE/csrc/rasterize_points/rasterization_utils.cuh:36-42``) and its unused
``get_ray_camera_space`` (``VoGE/Aggregation.py:11-27``, same signs, no half-pixel shift):

* row-vector convention ``X_view = X_world @ R + T``; view axes +X left, +Y up, +Z forward;
* pixel (row i, col j) centre has view direction ``[(px-j-0.5)/fx, (py-i-0.5)/fy, 1]``;
* world direction = normalise(dir_view @ R^-1); camera centre ``C = -T @ R^-1``.
"""
import numpy as np


def _pair(v):
    v = np.asarray(v, dtype=np.float64)
    if v.ndim == 0:
        return np.array([[float(v), float(v)]])
    if v.ndim == 1:
        return v[None, :] if v.shape[0] == 2 else np.stack([v, v], -1)
    return v


def pixel_rays(R, T, focal, principal, image_size):
    """R [B,3,3], T [B,3], focal scalar|(fx,fy)|[B,2], principal (px,py)|[B,2], image_size (H,W).

    Returns unit world-space directions [B,H,W,3] float32 and camera centres [B,3] float64."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    T = np.asarray(T, dtype=np.float64).reshape(-1, 3)
    B = R.shape[0]
    f = np.broadcast_to(_pair(focal), (B, 2))
    pp = np.broadcast_to(_pair(principal), (B, 2))
    H, W = int(image_size[0]), int(image_size[1])
    i = np.arange(H, dtype=np.float64)[:, None]
    j = np.arange(W, dtype=np.float64)[None, :]
    out = np.empty((B, H, W, 3))
    cen = np.empty((B, 3))
    for b in range(B):
        dv = np.stack(np.broadcast_arrays((pp[b, 0] - j - 0.5) / f[b, 0],
                                          (pp[b, 1] - i - 0.5) / f[b, 1],
                                          np.ones((H, W))), -1)
        Rinv = np.linalg.inv(R[b])
        dw = dv @ Rinv
        out[b] = dw / np.linalg.norm(dw, axis=-1, keepdims=True)
        cen[b] = -T[b] @ Rinv
    return out.astype(np.float32), cen


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, at=(0, 0, 0), up=(0, 1, 0)):
    """PyTorch3D's spherical look-at: C = dist*[cos e sin a, sin e, cos e cos a] (+at);
    z = norm(at-C), x = norm(up x z), y = norm(z x x); R has x,y,z as columns; T = -R^T C."""
    dist, elev, azim = np.broadcast_arrays(*(np.atleast_1d(np.asarray(v, np.float64)) for v in (dist, elev, azim)))
    if degrees:
        elev, azim = np.deg2rad(elev), np.deg2rad(azim)
    at = np.asarray(at, np.float64).reshape(-1, 3)
    up = np.broadcast_to(np.asarray(up, np.float64).reshape(-1, 3), (dist.shape[0], 3))
    C = np.stack([dist * np.cos(elev) * np.sin(azim), dist * np.sin(elev), dist * np.cos(elev) * np.cos(azim)], -1) + at

    def nrm(v):
        return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-5)
    z = nrm(at - C)
    x = nrm(np.cross(up, z))
    y = nrm(np.cross(z, x))
    bad = np.all(np.isclose(x, 0, atol=5e-3), axis=-1)
    if bad.any():
        x[bad] = nrm(np.cross(y, z))[bad]
    R = np.stack([x, y, z], axis=-1)  # columns
    T = -np.einsum('bij,bi->bj', R, C)
    return R.astype(np.float32), T.astype(np.float32)


def look_at_rotation(camera_position, at=(0, 0, 0), up=(0, 1, 0)):
    """PyTorch3D's look_at_rotation (same published convention as look_at_view_transform above): R with columns
    x = norm(up x z), y = norm(z x x), z = norm(at - C); a degenerate x (up parallel to z) is replaced by norm(y x z).
    Restated from the documentation: PyTorch3D is absent here, so this row stays "parity unpinned"."""
    C = np.asarray(camera_position, np.float64).reshape(-1, 3)
    at = np.broadcast_to(np.asarray(at, np.float64).reshape(-1, 3), C.shape)
    up = np.broadcast_to(np.asarray(up, np.float64).reshape(-1, 3), C.shape)

    def nrm(v):
        return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-5)
    z = nrm(at - C)
    x = nrm(np.cross(up, z))
    y = nrm(np.cross(z, x))
    bad = np.all(np.isclose(x, 0, atol=5e-3), axis=-1)
    if bad.any():
        x[bad] = nrm(np.cross(y, z))[bad]
    return np.stack([x, y, z], axis=-1)


def expand_sigma(sigma):
    """Aggregation.py:144-175 without a rotation: (N,)->s*I, (N,3)->diag, (N,3,3) passthrough."""
    sigma = np.asarray(sigma)
    if sigma.ndim == 3:
        assert sigma.shape[1:] == (3, 3)
        return sigma
    eye = np.eye(3, dtype=sigma.dtype)
    if sigma.ndim == 1:
        return sigma[:, None, None] * eye[None]
    if sigma.ndim == 2:
        return sigma[:, :, None] * eye[None]
    raise ValueError(sigma.shape)
