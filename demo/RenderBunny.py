"""Counterpart of the reference's demo/RenderBunny.py (BASELINE config 2): the Stanford bunny as 8171
isotropic Gaussians (`naive_vertices_converter(percentage=0.6)` of data/bunny.off, RenderBunny.py:17-24), colours
= vertex normals * 0.4 + 0.4 (:29), f = 2000, 256 x 256, max_assign = 40, look_at(6, 0, 10) (:27-41).

The mesh file and PyTorch3D's normals are not available here: the Gaussians and colours are the fixture
tests/golden/bunny_gaussians.npz, which tests/golden/make_golden.py produced with the reference's own
loader and converter.  With `--off path/to/bunny.off` the file is loaded and converted by this package's
IO / Converters instead (area-weighted vertex normals computed here).

usage: python demo/RenderBunny.py [--off FILE] [--out PREFIX] [--turntable N]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from VoGE.Converter.Converters import naive_vertices_converter                         # noqa: E402
from VoGE.Converter.IO import load_off                                                 # noqa: E402
from VoGE.Meshes import GaussianMeshesNaive                                            # noqa: E402
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background   # noqa: E402
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform               # noqa: E402


def vertex_normals(verts, faces):
    v, f = np.asarray(verts, np.float64), np.asarray(faces, np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])      # area-weighted face normals
    n = np.zeros_like(v)
    for c in range(3):
        np.add.at(n, f[:, c], fn)
    return n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-12)


ap = argparse.ArgumentParser()
ap.add_argument("--off", default=None)
ap.add_argument("--out", default="bunny")
ap.add_argument("--turntable", type=int, default=0, help="also time N views around the object")
a = ap.parse_args()
device = "cuda:0"
if a.off:
    verts_, faces_ = load_off(a.off)
    v, s, r = naive_vertices_converter(np.asarray(verts_), np.asarray(faces_), percentage=0.6)
    verts, sigmas = torch.as_tensor(np.asarray(v), dtype=torch.float32), torch.as_tensor(np.asarray(s), dtype=torch.float32)
    color = torch.as_tensor(vertex_normals(verts_, faces_) * 0.4 + 0.4, dtype=torch.float32)
else:
    g = np.load(os.path.join(ROOT, "tests", "golden", "bunny_gaussians.npz"))
    verts, sigmas, color = (torch.from_numpy(g[k]) for k in ("verts", "isigma", "colors"))
meshes = GaussianMeshesNaive(verts, sigmas, None).to(device)
color = color.to(device)

render_settings = GaussianRenderSettings(batch_size=-1, image_size=(256, 256), max_assign=40, absorptivity=1, principal=(128, 128),
                                         inverse_sigma=False)
cameras = PerspectiveCameras(focal_length=2000.0, principal_point=((128, 128),), image_size=(render_settings['image_size'],),
                             device=device, in_ndc=False)
renderer = GaussianRenderer(cameras=cameras, render_settings=render_settings)
R, T = look_at_view_transform([6], [0], [10], degrees=True)
cameras.R = R.to(device)
cameras.T = T.to(device)
with torch.no_grad():
    frag = renderer(meshes)
    img = to_white_background(frag.copy(), color).squeeze(0)
arr = img.cpu().numpy()
np.save(a.out + ".npy", arr)
print("image", arr.shape, "max", float(arr.max()), "covered pixels", int((frag.valid_num > 0).sum()), "->", a.out + ".npy")
try:
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.imsave(a.out + ".png", np.clip(arr, 0, 1))
    print("->", a.out + ".png")
except ImportError:
    pass
if a.turntable > 0:
    azim = [10 + 360.0 * i / a.turntable for i in range(a.turntable)]
    Rs, Ts = look_at_view_transform([6] * a.turntable, [0] * a.turntable, azim, degrees=True, device=device)   # all poses at once
    with torch.no_grad():
        for i in range(3):
            to_white_background(renderer(meshes, R=Rs[i:i + 1], T=Ts[i:i + 1]), color)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.turntable):
            to_white_background(renderer(meshes, R=Rs[i:i + 1], T=Ts[i:i + 1]), color)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.turntable
    print(f"turntable: {dt * 1e3:.3f} ms per view ({1 / dt:.0f} views/s, forward only, eager launches)")
