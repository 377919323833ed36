"""Per-rank compute of the pixel-row sharding, measured on ONE GPU: forward+backward of a band of H/n
rows (HIP graph replay, no collectives) for n = 1, 2, 4, 8 -- the part of strong scaling that does
not depend on xGMI.  With VIEWS=B a batch of B views is sharded on the stacked (view, row) axis instead
(distributed.stacked_bounds: whole views per rank when B % n == 0).
STRIPE=h deals the frame in interleaved stripes of h rows instead of contiguous bands (distributed.Stripes; round 4).
usage: [VIEWS=8 | STRIPE=32] python tools/band_time.py [config]"""
import os, sys, time, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from voge_amd.distributed import Stripes, balanced_row_bounds, projected_row_weight, render_stacked, row_band, stacked_bounds, stripe_height
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
VIEWS = int(os.environ.get("VIEWS", "1"))
R, T = look_at_view_transform(dist=[dd] * VIEWS, elev=[el] * VIEWS, azim=[az + 360.0 / VIEWS * b for b in range(VIEWS)], device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
base = None
FLOOR = float(os.environ.get("FLOOR", "-1"))       # >= 0: work-balanced bands (balanced_row_bounds) with this floor
STRIPE = int(os.environ.get("STRIPE", "0"))        # > 0: interleaved stripes of (at most) this many rows
weight = projected_row_weight(verts, R[0], T[0], focal, pp[1], H)
for n in (1, 2, 4, 8):
    worst = 0.0
    per = []
    for r in range(n):
        rows = row_band(H, r, n)
        if FLOOR >= 0:
            b = balanced_row_bounds(weight, n, floor=FLOOR)
            rows = (b[r], b[r + 1])
        if STRIPE > 0 and n > 1:
            rows = Stripes(H, r, n, stripe_height(H, n, STRIPE))
        def step():
            for p in params: p.grad = None
            if VIEWS == 1:
                to_white_background(renderer(gm, R=R, T=T, rows=rows), colors).sum().backward()
            else:
                sb = stacked_bounds(VIEWS, H, n)
                render_stacked(lambda b0, b1, r0, r1: to_white_background(renderer(gm, R=R[b0:b1], T=T[b0:b1], rows=(r0, r1)),
                                                                          colors.repeat(b1 - b0, 1)), sb[r], sb[r + 1], H).sum().backward()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): step()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): step()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        per.append(dt * 1e6)
    worst = max(per)
    base = base or worst
    print(f"n={n}: band us per rank {[round(x) for x in per]}  slowest {worst:.0f} us -> compute-only speedup {base / worst:.2f}x (efficiency {base / worst / n:.2f})")
