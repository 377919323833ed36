"""Mesh / point-cloud -> isotropic Gaussian converters (VoGE/Converter/Converters.py:10-139).

The isotropic rule everywhere: a vertex whose neighbours are at mean distance l gets
sigma = l^2 / (2 ln(1/percentage)) + 1e-10 and the renderer's input is 1/sigma."""
import numpy as np
import torch

from ..Meshes import GaussianMeshes


def get_vert_edge_length(verts, faces, default_l=1e-3):
    """Mean distance from every vertex to the distinct vertices it shares a face with
    (Converters.py:10-32); vertices in no face get default_l."""
    verts = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces)[:, :3].astype(np.int64)
    pairs = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [0, 2]]], axis=0)
    pairs = np.unique(np.sort(pairs, axis=1), axis=0)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    d = np.linalg.norm(verts[pairs[:, 0]] - verts[pairs[:, 1]], axis=1)
    total = np.zeros(len(verts))
    deg = np.zeros(len(verts))
    for col in (0, 1):
        np.add.at(total, pairs[:, col], d)
        np.add.at(deg, pairs[:, col], 1)
    out = np.full(len(verts), float(default_l))
    touched = np.zeros(len(verts), bool)
    touched[faces.ravel()] = True
    has = deg > 0
    out[has] = total[has] / deg[has]
    out[touched & ~has] = np.nan   # degenerate face (all three indices equal): 0/0 in the reference too
    return out


def _iso_from_length(length, percentage):
    return 1.0 / (length ** 2 / (2 * np.log(1 / percentage)) + 1e-10)


def naive_vertices_converter(vertices, faces, percentage=0.5, max_sig_rate=-1):
    """One isotropic Gaussian per mesh vertex (Converters.py:74-95) -> (verts, isigma, None)."""
    is_torch = torch.is_tensor(vertices)
    if is_torch:
        vertices, faces = vertices.numpy(), faces.numpy()
    default_l = 10 * np.sum((vertices.max(axis=0) - vertices.min(axis=0)) ** 2) ** 0.5 / vertices.shape[0]
    isigma = _iso_from_length(get_vert_edge_length(vertices, faces, default_l), percentage)
    if max_sig_rate > 0:
        isigma = np.minimum(isigma, np.mean(isigma) * max_sig_rate)
    if is_torch:
        return torch.from_numpy(vertices).type(torch.float32), torch.from_numpy(isigma).type(torch.float32), None
    return vertices, isigma, None


def normal_mesh_converter(vertices, faces, normals, percentage=0.5, shape_ratio=0.5, max_sig_rate=-1, auto_fix=True):
    """One Gaussian per mesh vertex, flattened along the vertex normal (Converters.py:35-71): in the frame whose third
    axis is the normal, Sigma^-1 = s * diag(1, 1, shape_ratio) with s the isotropic scale of naive_vertices_converter;
    the frame is look_at_rotation(-normal) (third column = normal; PyTorch3D's convention, cameras.look_at_rotation
    here).  Returns (verts, isigma [n,3,3], None)."""
    from ..cameras import look_at_rotation
    is_torch = torch.is_tensor(vertices)
    if is_torch:
        vertices, faces = vertices.numpy(), faces.numpy()
    else:
        normals = torch.from_numpy(np.asarray(normals))
    default_l = 10 * np.sum((vertices.max(axis=0) - vertices.min(axis=0)) ** 2) ** 0.5 / vertices.shape[0]
    base = _iso_from_length(get_vert_edge_length(vertices, faces, default_l), percentage)
    n2 = (normals ** 2).sum(-1)
    assert torch.max(n2) < 1.1 and torch.min(n2) > 0.9
    shape = np.array([[1, 0, 0], [0, 1, 0], [0, 0, shape_ratio]])[None] * base.reshape(-1, 1, 1)
    rot = look_at_rotation(-normals.type(torch.float32)).numpy()
    isigma = rot @ shape @ rot.transpose(0, 2, 1)
    if auto_fix:
        flat = np.linalg.det(isigma) == 0
        isigma[flat] = np.eye(3)[None] * base[flat].reshape(-1, 1, 1)
    if max_sig_rate > 0:
        isigma = np.minimum(isigma, np.mean(isigma) * max_sig_rate)
    if is_torch:
        return torch.from_numpy(vertices).type(torch.float32), torch.from_numpy(isigma).type(torch.float32), None
    return vertices, isigma, None


def fixed_pointcloud_converter(points, radius, percentage=0.5):
    """Isotropic Gaussians of a given radius per point (Converters.py:125-139)."""
    to_np = not torch.is_tensor(points)
    if to_np:
        points = torch.from_numpy(np.asarray(points))
        if not isinstance(radius, float):
            radius = torch.from_numpy(np.asarray(radius))
    isigma = torch.ones(points.shape[0]) / ((radius ** 2) / (2 * np.log(1 / percentage)) + 1e-10)
    return (points.numpy(), isigma.numpy(), None) if to_np else (points, isigma, None)


def naive_point_cloud_converter(points, percentage=0.5, n_nearest=4, thr_max=2, chunk=4096):
    """Isotropic Gaussians from the mean distance to the n nearest neighbours, each neighbour
    distance capped at thr_max x their mean (Converters.py:98-122; note the 4 ln(1/p) divisor)."""
    to_np = not torch.is_tensor(points)
    pts = torch.as_tensor(points).type(torch.float32)
    out = []
    with torch.no_grad():
        for s in range(0, pts.shape[0], chunk):
            dist = torch.cdist(pts[s:s + chunk], pts)
            top = torch.topk(dist, k=n_nearest, dim=1, largest=False)[0]
            length = torch.min(top, top.mean(dim=1, keepdim=True) * thr_max).mean(dim=1)
            out.append(length ** 2 / (4 * np.log(1 / percentage)))
    isigma = 1 / (torch.cat(out) + 1e-8)
    return (pts.numpy(), isigma.numpy(), None) if to_np else (pts, isigma, None)


def to_gaussian_meshes(converter, **kwargs):
    """converter(verts, faces, **kwargs) -> GaussianMeshes factory taking (verts, faces) tensors; the
    PyTorch3D-free counterpart of pytorch3d2gaussian (Converters.py:176-194)."""
    def wrapper(verts, faces=None, device="cpu", **mesh_kwargs):
        args = (verts.cpu(), faces.cpu()) if faces is not None else (verts.cpu(),)
        v, s, r = converter(*args, **kwargs)
        return GaussianMeshes(v.type(torch.float32), s.type(torch.float32), None if r is None else r.type(torch.float32),
                              **mesh_kwargs).to(device)
    return wrapper
