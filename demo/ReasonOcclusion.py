"""Occlusion reasoning by gradient descent: the counterpart of the reference's demo/ReasonOcclusion.py.

Two translucent cuboids (surface Gaussians from `Cuboid.cuboid_gauss`, 4000 + 3000 requested, percentage 0.7;
ReasonOcclusion.py:25-37), the first in front of the second at the target pose (offsets (0.5, 0, 1) and (0, 0, 0), :39-40).
The optimisation starts with the first cuboid BEHIND the second ((-1, 0, -5), :79) and moves both translations with Adam
(lr 0.05, betas (0.6, 0.4), :85) on the MSE between `interpolate_attr` images (:101-105) for 200 iterations: the gradient has
to carry the cuboid THROUGH the other one, which only a renderer with volume (depth-ordered, translucent) contributions can
do.  Settings of :51-55: 400 x 400, focal 300, max_assign 60, max_point_per_bin 1500; camera dist 5, elev 10, azim 20 (:42).

What differs: no PNG / MP4 side outputs unless --save is given (PIL only); the run reports, and returns, the translations
and the loss history; `tests/test_gpu_demo_loops.py` asserts convergence (both translations within 0.05 of the target).

usage: python demo/ReasonOcclusion.py [--iters 200] [--save DIR]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from VoGE.Converter import Cuboid                                                                      # noqa: E402
from VoGE.Meshes import GaussianMeshesNaive                                                            # noqa: E402
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, interpolate_attr, to_white_background  # noqa: E402
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform                               # noqa: E402

FACE_COLORS = (np.array([[0, 0.2, 1], [0, 0.2, 1], [0, 1, 0.2], [0, 1, 0.2], [0, 1, 1], [0, 1, 1]]),          # :25
               np.array([[1, 0.2, 0], [1, 0.2, 0], [1, 1, 0], [1, 1, 0], [0.2, 1, 0], [0.2, 1, 0]]))          # :32
EXTENTS = (((-0.8, 0.8), (-0.4, 0.4), (-0.6, 0.6), 4000), ((-1, 1), (-1, 1), (-0.3, 0.3), 3000))               # :26, :33
TARGET = ((0.5, 0.0, 1.0), (0.0, 0.0, 0.0))                                                                    # :39-40
START = ((-1.0, 0.0, -5.0), (0.0, 0.0, 0.0))                                                                   # :79-80


def build_scene(device, percentage=0.7):
    parts = []
    for (xr, yr, zr, n), cols in zip(EXTENTS, FACE_COLORS):
        v, s, c = Cuboid.cuboid_gauss(xr, yr, zr, n, colors=cols, percentage=percentage)
        parts.append(tuple(torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device) for a in (v, s, c)))
    return parts


def save_png(img, path):
    from PIL import Image
    a = img.clamp(min=0, max=1)[0, ..., :3].detach().cpu().numpy() * 255
    Image.fromarray(a.astype(np.uint8)).save(path)


def run(iters=200, device="cuda", save=None, log=print):
    """-> {"loss": [...], "v0": final translation of the first cuboid, "v1": of the second, "sec_per_iter": s}"""
    size = (400, 400)
    (v0, s0, c0), (v1, s1, c1) = build_scene(device)
    sig, cols = torch.cat((s0, s1)), torch.cat((c0, c1))
    R, T = look_at_view_transform(dist=5, elev=10, azim=20, device=device)
    cams = PerspectiveCameras(focal_length=300, principal_point=((size[0] // 2, size[1] // 2),), image_size=(size,), device=device,
                              R=R, T=T)
    settings = GaussianRenderSettings(max_assign=60, principal=(size[0] // 2, size[1] // 2), image_size=size, max_point_per_bin=1500)
    renderer = GaussianRenderer(cameras=cams, render_settings=settings).to(device)

    def place(t0, t1):
        return GaussianMeshesNaive(verts=torch.cat((v0 + t0, v1 + t1), dim=0), sigmas=sig)

    tgt0, tgt1 = (torch.tensor([t], dtype=torch.float32, device=device) for t in TARGET)
    with torch.no_grad():
        target = interpolate_attr(renderer(place(tgt0, tgt1), R=R, T=T), cols)
    p0, p1 = (torch.tensor([t], dtype=torch.float32, device=device, requires_grad=True) for t in START)
    opt = torch.optim.Adam([p0, p1], lr=0.05, betas=(0.6, 0.4))
    if save:
        os.makedirs(save, exist_ok=True)
        with torch.no_grad():
            save_png(target, os.path.join(save, "target.png"))
            save_png(to_white_background(renderer(place(p0, p1), R=R, T=T), cols), os.path.join(save, "before.png"))
    losses = []
    warm = min(10, iters // 2)      # (the first iterations load the library and make the first allocations: not timed)
    t_start = time.perf_counter()
    for it in range(iters):
        if it == warm:
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        img = interpolate_attr(renderer(place(p0, p1), R=R, T=T), cols)
        loss = torch.nn.functional.mse_loss(img, target)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t_start) / max(iters - warm, 1)
    losses = [float(x) for x in losses]
    out = {"loss": losses, "v0": p0.detach().cpu().numpy()[0], "v1": p1.detach().cpu().numpy()[0], "sec_per_iter": sec}
    log(f"{iters} iterations, {sec * 1e3:.2f} ms each: loss {losses[0]:.5f} -> {losses[-1]:.2e}; "
        f"first cuboid {np.round(out['v0'], 3)} (target {TARGET[0]}), second {np.round(out['v1'], 3)} (target {TARGET[1]})")
    if save:
        with torch.no_grad():
            save_png(to_white_background(renderer(place(p0, p1), R=R, T=T), cols), os.path.join(save, "after.png"))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    run(a.iters, save=a.save)
