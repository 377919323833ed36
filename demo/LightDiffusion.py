"""Diffuse shading from interpolated normals: the counterpart of the reference's demo/LightDiffusion.py.

The Stanford bunny as 8171 isotropic Gaussians (`naive_vertices_converter(percentage=0.6)`, LightDiffusion.py:28), its
per-vertex normals as the attribute; `interpolate_attr` gives a normal per pixel (:57) and a directional light shades it.
The reference takes the shading from PyTorch3D (`DirectionalLights.diffuse`, :58-59): colour * relu(<n, l>) with both
vectors normalised -- restated here in three lines of torch.  Settings of :32-47: 256 x 256, focal 2000, max_assign 40,
camera (dist 6, elev 0, azim 10); light from (elev 30 + |100 - k| / 2, azim 10), the reference's single frame is k = 5.

Data: tests/golden/bunny_gaussians.npz (vertices, converter output and faces of data/bunny.off, made by
tests/golden/make_golden.py with the reference's loader and converter); `--off FILE` loads a mesh with this package's IO.

usage: python demo/LightDiffusion.py [--off FILE] [--frames 1] [--out PREFIX]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from VoGE.Converter.Converters import naive_vertices_converter                          # noqa: E402
from VoGE.Converter.IO import load_off                                                 # noqa: E402
from VoGE.Meshes import GaussianMeshesNaive                                            # noqa: E402
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, interpolate_attr   # noqa: E402
from voge_amd.cameras import PerspectiveCameras, camera_position_from_spherical_angles, look_at_view_transform  # noqa: E402


def vertex_normals(verts, faces):
    """Unit vertex normals: the area-weighted sum of the adjacent faces' normals (what Meshes.verts_normals_packed gives)."""
    v, f = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for c in range(3):
        np.add.at(n, f[:, c], fn)
    return (n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-12)).astype(np.float32)


def diffuse(normals_map, direction, color):
    """colour * max(<n, l>, 0) with n and l normalised (the directional light's diffuse term)."""
    n = torch.nn.functional.normalize(normals_map, p=2, dim=-1, eps=1e-6)
    l = torch.nn.functional.normalize(direction, p=2, dim=-1, eps=1e-6)
    return color * torch.relu((n * l).sum(-1))[..., None]


def run(off=None, frames=1, out=None, device="cuda:0", log=print):
    if off:
        v_, f_ = load_off(off)
        v, s, _ = naive_vertices_converter(np.asarray(v_), np.asarray(f_), percentage=0.6)
        faces = np.asarray(f_)
    else:
        g = np.load(os.path.join(ROOT, "tests", "golden", "bunny_gaussians.npz"))
        v, s, faces = g["verts"], g["isigma"], g["faces"]
    normals = torch.from_numpy(vertex_normals(v, faces)).to(device)
    meshes = GaussianMeshesNaive(torch.as_tensor(np.asarray(v), dtype=torch.float32), torch.as_tensor(np.asarray(s), dtype=torch.float32)).to(device)
    settings = GaussianRenderSettings(batch_size=-1, image_size=(256, 256), max_assign=40, absorptivity=1, principal=(128, 128),
                                      inverse_sigma=False)
    cameras = PerspectiveCameras(focal_length=2000.0, principal_point=((128, 128),), image_size=(settings['image_size'],),
                                 device=device, in_ndc=False)
    renderer = GaussianRenderer(cameras=cameras, render_settings=settings)
    R, T = look_at_view_transform([6], [0], [10], degrees=True)
    cameras.R, cameras.T = R.to(device), T.to(device)
    color = torch.ones((1, 3), device=device)
    images = []
    with torch.no_grad():
        frag = renderer(meshes)
        normals_map = interpolate_attr(frag, normals)                 # [1, 256, 256, 3]: one pass, reused by every light
        ks = [5] if frames == 1 else list(np.linspace(0, 200, frames))
        for k in ks:
            direction = camera_position_from_spherical_angles(1, 30 + abs(100 - k) * 0.5, 10, device=device)
            images.append(diffuse(normals_map, direction, color))
    img = torch.cat(images, 0)
    lit = float((img[0].sum(-1) > 0).float().mean())
    log(f"{len(ks)} light direction(s): image {tuple(img.shape)}, lit pixels {lit * 100:.1f} %, max {float(img.max()):.3f}")
    if out:
        np.save(out + ".npy", img.cpu().numpy())
        try:
            from PIL import Image
            for i in range(img.shape[0]):
                Image.fromarray((img[i].clamp(0, 1) * 255).cpu().numpy().astype(np.uint8)).save(f"{out}_{i:03d}.png")
        except ImportError:
            pass
    return {"image": img, "normals_map": normals_map, "frag": frag, "normals": normals}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--off", default=None)
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--out", default="light")
    a = ap.parse_args()
    run(a.off, a.frames, a.out)
