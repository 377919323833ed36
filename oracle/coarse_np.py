"""numpy restatement of the reference's COARSE stage -- TEST INFRASTRUCTURE (see oracle/__init__.py).

What it follows:
  VoGE/RayTracing.py:33-73      convert_to_box, rasterize_coarse (the projection in front of the kernel)
  VoGE/csrc/rasterize_coarse/rasterize_coarse.cu:20-42    EllipseBoundingBoxKernel (bbox = centre -+ radius, skip z < 0)
  .../rasterize_coarse.cu:44-188                          RasterizeCoarseCudaKernel (bbox / bin overlap with the half-pixel
                                                          pad, 512-element chunks, a chunk that does not fit is DROPPED)
  VoGE/csrc/rasterize_points/rasterization_utils.cuh:15-42 NonSquareNdcRange, PixToNonSquareNdc

The projection goes through PyTorch3D (un-vendored, absent): PerspectiveCameras in screen space,
get_full_projection_transform().compose(get_ndc_camera_transform()), restated here from its documentation
([recall]) -- that part is "parity unpinned" like the ray convention.  The kernel part is
restated statement by statement.  Two things the CUDA kernel leaves to the scheduler are fixed here: chunks are
taken in ascending order (the reference's order between chunks depends on which block's atomicAdd lands first),
so a bin's list is ascending in index and, when a bin overflows, it is the LATER chunks that are dropped.
"""
import numpy as np


def ndc_range(s1, s2):
    """NonSquareNdcRange (rasterization_utils.cuh:15-23), fp32."""
    r = np.float32(2.0)
    if s1 > s2:
        r = np.float32(s1 * r) / np.float32(s2)
    return np.float32(r)


def pix_to_ndc(i, s1, s2):
    """PixToNonSquareNdc (rasterization_utils.cuh:36-42), fp32, same operation order."""
    r = ndc_range(s1, s2)
    off = np.float32(r / np.float32(2.0))
    return np.float32(-off + np.float32(np.float32(r * np.float32(i)) + off) / np.float32(s1))


def project_for_coarse(mus, isg, R, T, focal, principal, image_size, thr):
    """RayTracing.py:42-57 + convert_to_box (:33-39).  mus [B,N,3] camera-centred means, isg [B,N,3,3] = A,
    R [B,3,3], T [B,3], focal / principal in pixels.  Returns points [B,N,3] f32 = (x_ndc, y_ndc, view z) and
    radius [B,N,2] f32 (NaN where the column sums of the projected 2x2 block are negative, as in the reference)."""
    mus = np.asarray(mus, np.float64)
    A = np.asarray(isg, np.float64)
    R = np.asarray(R, np.float64).reshape(-1, 3, 3)
    T = np.asarray(T, np.float64).reshape(-1, 3)
    B = mus.shape[0]
    H, W = int(image_size[0]), int(image_size[1])
    f = np.broadcast_to(np.atleast_1d(np.asarray(focal, np.float64)).reshape(-1, 1) if np.ndim(focal) < 2 else np.asarray(focal, np.float64), (B, 2)) \
        if np.ndim(focal) >= 1 else np.full((B, 2), float(focal))
    pp = np.broadcast_to(np.asarray(principal, np.float64).reshape(-1, 2), (B, 2))
    s = float(min(H, W))
    pts = np.empty((B, mus.shape[1], 3), np.float32)
    rad = np.empty((B, mus.shape[1], 2), np.float32)
    for b in range(B):
        C = -np.linalg.inv(R[b].T) @ T[b]                       # :45  (the camera centre with the opposite sign)
        world = mus[b] + C[None]                                # :46
        view = world @ R[b] + T[b][None]                        # X_view = X_world R + T
        z = view[:, 2]
        # image coordinates (x right, y down) of the projection: view axes are +X left / +Y up, and the pixel that looks
        # along (X/Z, Y/Z) is column px - fx X/Z, row py - fy Y/Z (the ray convention of oracle/camera_np.py; the
        # reference must agree with its own ray sampler here or its default path would bin every Gaussian mirrored)
        xs = pp[b, 0] - f[b, 0] * view[:, 0] / z
        ys = pp[b, 1] - f[b, 1] * view[:, 1] / z
        # image -> NDC (+X left) is (W/2 - xs) 2/s; the reference negates it (:50): +x right, pixel 0 at -range/2
        pts[b, :, 0] = ((xs - W / 2.0) * 2.0 / s)
        pts[b, :, 1] = ((ys - H / 2.0) * 2.0 / s)
        pts[b, :, 2] = z                                        # :57 (view-space depth replaces the NDC z)
        Av = np.einsum("ji,njk,kl->nil", R[b], A[b], R[b])      # R^T A R  (:52-53)
        F = np.diag([-2.0 * f[b, 0] / s, -2.0 * f[b, 1] / s])
        get = -np.log(thr) * (F[None] @ np.linalg.inv(Av[:, :2, :2]) @ F[None])      # :35
        with np.errstate(invalid="ignore"):
            rad[b] = (np.sqrt(get.sum(axis=1)) * (1.0 / z)[:, None])             # ones(1,2) @ get = column sums; * z = 1/Z (:38)
    return pts, rad


def rasterize_points_coarse(points, first_idx, num_points, image_size, radius, bin_size, max_points_per_bin, chunk_size=512):
    """points [P,3] f32 (x, y, z), radius [P,2] f32, first_idx / num_points [B] int64 -> bin_elems [B,BH,BW,M] int32."""
    points = np.asarray(points, np.float32)
    radius = np.asarray(radius, np.float32)
    H, W = int(image_size[0]), int(image_size[1])
    M = int(max_points_per_bin)
    nbx, nby = 1 + (W - 1) // bin_size, 1 + (H - 1) // bin_size
    assert nbx < 66 and nby < 66, "kMaxItemsPerBin (rasterize_coarse.cu:213)"
    B = len(first_idx)
    P = points.shape[0]
    xmin, xmax = points[:, 0] - radius[:, 0], points[:, 0] + radius[:, 0]
    ymin, ymax = points[:, 1] - radius[:, 1], points[:, 1] + radius[:, 1]
    skip = points[:, 2] < 0
    half_x = np.float32(ndc_range(W, H) / np.float32(2.0)) / np.float32(W)
    half_y = np.float32(ndc_range(H, W) / np.float32(2.0)) / np.float32(H)
    bx_min = np.array([pix_to_ndc(b * bin_size, W, H) - half_x for b in range(nbx)], np.float32)
    bx_max = np.array([pix_to_ndc((b + 1) * bin_size - 1, W, H) + half_x for b in range(nbx)], np.float32)
    by_min = np.array([pix_to_ndc(b * bin_size, H, W) - half_y for b in range(nby)], np.float32)
    by_max = np.array([pix_to_ndc((b + 1) * bin_size - 1, H, W) + half_y for b in range(nby)], np.float32)
    out = np.full((B, nby, nbx, M), -1, np.int32)
    count = np.zeros((B, nby, nbx), np.int64)
    chunks_per_batch = 1 + (P - 1) // chunk_size if P > 0 else 0
    for b in range(B):
        e0, e1 = int(first_idx[b]), int(first_idx[b]) + int(num_points[b])
        for c in range(chunks_per_batch):
            lo = c * chunk_size
            e = np.arange(lo, min(lo + chunk_size, P))
            e = e[(e >= e0) & (e < e1) & ~skip[e]]
            if e.size == 0:
                continue
            with np.errstate(invalid="ignore"):
                oy = (ymin[e][:, None] <= by_max[None]) & (by_min[None] < ymax[e][:, None])      # [n, nby]
                ox = (xmin[e][:, None] <= bx_max[None]) & (bx_min[None] < xmax[e][:, None])      # [n, nbx]
            for y in range(nby):
                ey = e[oy[:, y]]
                if ey.size == 0:
                    continue
                oxy = ox[oy[:, y]]
                for x in range(nbx):
                    ids = ey[oxy[:, x]]
                    if ids.size == 0:
                        continue
                    start = count[b, y, x]
                    count[b, y, x] += ids.size              # the counter moves even when the chunk is dropped (:154-170)
                    if start + ids.size > M:
                        continue
                    out[b, y, x, start:start + ids.size] = ids
    return out


def reference_candidate_lists(mus, isg, R, T, focal, principal, image_size, thr, n_assign, bin_size=None, max_points_per_bin=None):
    """ray_tracing's coarse branch (RayTracing.py:12-28): bin_size / max_points_per_bin defaults included."""
    B, N = np.asarray(mus).shape[:2]
    if bin_size is None:
        bin_size = max(int(2 ** np.ceil(np.log2(max(image_size)) - 5)), 10)
    if max_points_per_bin is None:
        max_points_per_bin = min(int(max(n_assign * 10, N / 10)), N)
    pts, rad = project_for_coarse(mus, isg, R, T, focal, principal, image_size, thr)
    first = np.arange(B, dtype=np.int64) * N
    num = np.full(B, N, np.int64)
    return rasterize_points_coarse(pts.reshape(-1, 3), first, num, image_size, rad.reshape(-1, 2), bin_size, max_points_per_bin), bin_size
