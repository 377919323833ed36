"""Timing scan for performance cliffs: the same scene seen from farther and farther away (the object shrinks on screen, the
per-tile candidate lists grow), several N and K.  One line per case: forward (trace only), frame forward+backward, hits per
pixel.  usage (GPU box): python tools/cliff_scan.py"""
import sys, time, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
dev = torch.device("cuda", 0)


def ev_time(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'N':>7s} {'HxW':>9s} {'K':>4s} {'dist':>5s} {'r':>11s} | {'trace us':>9s} {'frame us':>9s} {'hits/px':>8s} {'lit %':>6s}")
for N, size, K, (rlo, rhi) in ((2562, 128, 25, (0.04, 0.07)), (10000, 256, 40, (0.03, 0.06)), (50000, 512, 40, (0.02, 0.04)), (50000, 512, 25, (0.02, 0.04)),
                                (200000, 512, 40, (0.01, 0.02))):
    verts, sig, cols = scenes.random_gaussians(N, seed=0, r_lo=rlo, r_hi=rhi)
    for dist in (2.5, 4.0, 8.0, 16.0, 32.0):
        gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
        colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
        R, T = look_at_view_transform(dist=dist, elev=10.0, azim=70.0, device=dev)
        cams = PerspectiveCameras(focal_length=1.17 * size, principal_point=((size / 2.0, size / 2.0),), image_size=((size, size),), device=dev)
        renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(size, size), max_assign=K, max_point_per_bin=-1)).to(dev)
        with torch.no_grad():
            t_tr = ev_time(lambda: renderer(gm, R=R, T=T).vert_index)
            f = renderer(gm, R=R, T=T)
            hp = float(f.valid_num.float().mean()); lit = float((f.valid_num > 0).float().mean()) * 100

        def frame():
            for p in (gm.verts, gm.sigmas, colors):
                p.grad = None
            to_white_background(renderer(gm, R=R, T=T), colors).sum().backward()
        t_fr = ev_time(frame)
        print(f"{N:7d} {size:4d}x{size:<4d} {K:4d} {dist:5.1f} {rlo:.3f}-{rhi:.3f} | {t_tr:9.1f} {t_fr:9.1f} {hp:8.2f} {lit:6.1f}", flush=True)
