"""A coloured point cloud as fixed-radius Gaussians: the counterpart of the reference's demo/RenderPointClouds.py.

Settings of RenderPointClouds.py:31-45: `fixed_pointcloud_converter(radius=0.003, percentage=0.75)`, 320 x 320, focal 300,
the renderer's DEFAULT settings otherwise (max_assign 20, max_point_per_bin None -> the coarse stage's heuristic bins),
camera (dist 3.5, elev 10, azim 0), `to_white_background`.

What differs: the reference downloads PyTorch3D's PittsburghBridge cloud (438 544 points; :12-13); there is no network here,
so the default cloud is synthetic and of the same size and extent -- points on an undulating wall over a ground plane,
coloured by position (`--points N` changes its size) -- and `--npz FILE` renders any file with `verts` and `rgb` arrays as the
reference does (verts[:, 1] += 0.5, rgb * 0.85; :25-27).

usage: python demo/RenderPointClouds.py [--npz FILE] [--points 438544] [--out PREFIX]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from VoGE.Converter import Converters                                                    # noqa: E402
from VoGE.Meshes import GaussianMeshes                                                   # noqa: E402
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background   # noqa: E402
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform                  # noqa: E402


def synthetic_cloud(n, seed=0):
    """n points on an undulating wall standing on a ground plane (extent ~ [-1, 1] x [-0.5, 0.5] x [-0.6, 0.6]) and per-point
    colours."""
    rng = np.random.default_rng(seed)
    ground = rng.random(n) > 0.7
    x = rng.uniform(-1, 1, n)
    y = rng.uniform(-0.5, 0.5, n)
    z = 0.15 * np.sin(3 * x) * np.cos(4 * y) + 0.01 * np.sin(60 * x)            # the wall: relief with fine ridges
    y = np.where(ground, -0.5 + 0.01 * np.sin(25 * x), y)
    z = np.where(ground, rng.uniform(-0.6, 0.6, n), z)
    verts = np.stack((x, y, z), 1).astype(np.float32)
    rgb = np.stack((0.5 + 0.5 * x, 0.5 + y, 0.5 + 2.0 * z), 1).clip(0, 1).astype(np.float32)
    rgb[ground] = rgb[ground] * 0.4 + 0.3
    return verts, rgb


def run(npz=None, points=438544, out=None, device="cuda", log=print):
    if npz:
        pc = np.load(npz)
        verts = torch.tensor(pc["verts"], dtype=torch.float32)
        verts[:, 1] += 0.5
        rgb = torch.tensor(pc["rgb"][:, 0:3] * 0.85, dtype=torch.float32)
    else:
        v, c = synthetic_cloud(points)
        verts, rgb = torch.from_numpy(v), torch.from_numpy(c * 0.85)
    cameras = PerspectiveCameras(focal_length=300, principal_point=((160, 160),), image_size=((320, 320),), device=device, in_ndc=False)
    verts, sigmas, _ = Converters.fixed_pointcloud_converter(verts, radius=0.003, percentage=0.75)
    gmesh = GaussianMeshes(verts=verts, sigmas=sigmas).to(device)
    rgb = rgb.to(device)
    renderer = GaussianRenderer(cameras=cameras, render_settings=GaussianRenderSettings(image_size=(320, 320), principal_point=(160, 160)))
    R, T = look_at_view_transform(3.5, 10, 0, device=device)
    with torch.no_grad():
        for _ in range(2):
            frag = renderer(gmesh, R=R, T=T)
            img = to_white_background(frag, rgb).clamp(0, 1).squeeze(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frag = renderer(gmesh, R=R, T=T)
        img = to_white_background(frag, rgb).clamp(0, 1).squeeze(0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    covered = float((frag.valid_num > 0).float().mean())
    log(f"{verts.shape[0]} points, 320 x 320: {ms:.2f} ms per frame, {covered * 100:.1f} % of the pixels covered")
    if out:
        np.save(out + ".npy", img.cpu().numpy())
        try:
            from PIL import Image
            Image.fromarray((img * 255).cpu().numpy().astype(np.uint8)).save(out + ".png")
        except ImportError:
            pass
    return {"image": img, "frag": frag, "ms": ms}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--npz", default=None)
    ap.add_argument("--points", type=int, default=438544)
    ap.add_argument("--out", default="pointcloud")
    a = ap.parse_args()
    run(a.npz, a.points, a.out)
