"""Surface Gaussians of an axis-aligned cuboid: same vertex order, spacing rule and isotropic
Sigma^-1 scale as VoGE/Converter/Cuboid.py:8-67 (cuboid_gauss), vectorised; and the triangle mesh of
the same cuboid (cuboid_mesh, :70-159).

Spacing: each vertex owns ~total_area/(2*n) of surface, edge = sqrt(2 * that); axis samples are
linspace(lo, hi, int(extent/edge + 1)); sigma = edge^2 / (2 ln(1/percentage)) + 1e-10 and the
renderer's "sigmas" input is 1/sigma."""
import numpy as np
import torch

from ..Meshes import GaussianMeshes


def _axis(lo_hi, edge):
    lo, hi = lo_hi
    return lo + np.linspace(0, hi - lo, int((hi - lo) / edge + 1))


def cuboid_gauss(x_range, y_range, z_range, number_vertices, percentage=0.5, colors=None, as_obj=False):
    ext = [r[1] - r[0] for r in (x_range, y_range, z_range)]
    area = 2 * (ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2])
    edge = (2 * area / (number_vertices * 2)) ** 0.5
    xs, ys, zs = _axis(x_range, edge), _axis(y_range, edge), _axis(z_range, edge)

    def grid(a, b, c_fixed, order):
        """points with `a` fastest, `b` slowest, third coordinate fixed; `order` places (a,b,c) into xyz."""
        A, Bv = np.meshgrid(a, b)            # rows follow b, columns follow a
        pts = np.stack([A.ravel(), Bv.ravel(), np.full(A.size, c_fixed)], axis=1)
        return pts[:, order]

    zi = zs[1:-1]
    faces = [
        grid(xs, ys, zs[0], [0, 1, 2]),          # z = z_min : y rows, x fastest
        grid(xs, ys, zs[-1], [0, 1, 2]),         # z = z_max
        grid(xs[:-1], zi, ys[0], [0, 2, 1]),     # y = y_min : z rows, x fastest (last x left to the x_max face)
        grid(xs[1:], zi, ys[-1], [0, 2, 1]),     # y = y_max
        grid(ys[1:], zi, xs[0], [2, 0, 1]),      # x = x_min : z rows, y fastest
        grid(ys[:-1], zi, xs[-1], [2, 0, 1]),    # x = x_max
    ]
    verts = np.concatenate(faces, axis=0)
    isigma = 1.0 / (edge ** 2 / (2 * np.log(1 / percentage)) + 1e-10)
    sig = np.ones(len(verts)) * isigma
    out_colors = None
    if colors is not None:
        out_colors = np.concatenate([np.repeat(np.asarray(c)[None, :], len(f), axis=0) for f, c in zip(faces, colors)], axis=0)
    if as_obj:
        obj = GaussianMeshes(verts=torch.from_numpy(verts).type(torch.float32), sigmas=torch.from_numpy(sig).type(torch.float32))
        return obj if out_colors is None else (obj, out_colors)
    return (verts, sig) if out_colors is None else (verts, sig, out_colors)


def cuboid_mesh(x_range, y_range, z_range, number_vertices, colors=None, as_obj=False):
    """Triangulated surface of the cuboid (VoGE/Converter/Cuboid.py:70-159): six independent vertex grids (z_min,
    z_max, y_min, y_max, x_min, x_max -- edges and corners are duplicated, unlike cuboid_gauss), two triangles per
    grid cell.  Returns (verts [n,3] f64, faces [m,3] int[, colors]); as_obj=True wraps them in a PyTorch3D Meshes
    like the reference, which needs PyTorch3D."""
    ext = [r[1] - r[0] for r in (x_range, y_range, z_range)]
    area = 2 * (ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2])
    edge = (2 * area / (number_vertices * 2)) ** 0.5
    xs, ys, zs = _axis(x_range, edge), _axis(y_range, edge), _axis(z_range, edge)

    def grid(a, b, c_fixed, order):
        A, Bv = np.meshgrid(a, b)            # rows follow b, columns follow a
        return np.stack([A.ravel(), Bv.ravel(), np.full(A.size, c_fixed)], axis=1)[:, order]

    def cells(na, nb, base):
        """faces of an nb x na vertex grid (a fastest): (v, v+1, v+na) and (v+na+1, v+1, v+na) per cell."""
        m, n = np.meshgrid(np.arange(nb - 1), np.arange(na - 1), indexing="ij")
        v = (base + m * na + n).ravel()
        return np.stack([np.stack([v, v + 1, v + na], 1), np.stack([v + na + 1, v + 1, v + na], 1)], axis=1).reshape(-1, 3)

    sheets = [(xs, ys, zs[0], [0, 1, 2]), (xs, ys, zs[-1], [0, 1, 2]), (xs, zs, ys[0], [0, 2, 1]), (xs, zs, ys[-1], [0, 2, 1]),
              (ys, zs, xs[0], [2, 0, 1]), (ys, zs, xs[-1], [2, 0, 1])]
    verts, faces, base = [], [], 0
    for a, b2, c, order in sheets:
        verts.append(grid(a, b2, c, order))
        faces.append(cells(a.size, b2.size, base))
        base += a.size * b2.size
    out_v = np.concatenate(verts, axis=0)
    out_f = np.concatenate(faces, axis=0) if sum(len(f) for f in faces) else np.zeros((0, 3), np.int64)
    out_c = None
    if colors is not None:
        out_c = np.concatenate([np.repeat(np.asarray(c)[None, :], len(v), axis=0) for v, c in zip(verts, colors)], axis=0)
    if as_obj:
        try:
            from pytorch3d.structures import Meshes
        except ImportError as e:      # the reference returns a PyTorch3D object here (Cuboid.py:149-157)
            raise ImportError("cuboid_mesh(as_obj=True) returns a pytorch3d.structures.Meshes: PyTorch3D is not installed; "
                              "use as_obj=False for (verts, faces) arrays") from e
        mesh = Meshes(verts=[torch.from_numpy(out_v).type(torch.float32)], faces=[torch.from_numpy(out_f).type(torch.long)])
        return mesh if out_c is None else (mesh, out_c)
    return (out_v, out_f) if out_c is None else (out_v, out_f, out_c)
