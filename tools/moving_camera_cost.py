"""bench.py's eager_moving_camera variant, taken apart: the eager step's wall time per frame on fixed views of the same 64-view
sequence (GPU- or host-bound, whichever is larger), the GPU time of those views under graph replay, and the moving loop.
usage (GPU box): python tools/moving_camera_cost.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform

dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
nview = 64
Rm, Tm = look_at_view_transform(dist=[dd] * nview, elev=[el] * nview, azim=[az + 0.5 * i for i in range(nview)], device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
params = (gm.verts, gm.sigmas, colors)


def step(i):
    for p in params:
        p.grad = None
    to_white_background(renderer(gm, R=Rm[i:i + 1], T=Tm[i:i + 1]), colors).sum().backward()


def loop(fn, n=300):
    for k in range(60):
        fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        fn(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n


for i in (0, 16, 32, 48, 63):
    h, w = loop(lambda k: step(i))
    Ri, Ti = Rm[i:i + 1].clone(), Tm[i:i + 1].clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            for p in params:
                p.grad = None
            to_white_background(renderer(gm, R=Ri, T=Ti), colors).sum().backward()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for p in params:
                p.grad = None
            to_white_background(renderer(gm, R=Ri, T=Ti), colors).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    gus = 1e6 * (time.perf_counter() - t0) / 200
    print(f"view {i:2d} (azim {az + 0.5 * i:.1f}): eager loop host {h:.1f} us, wall {w:.1f} us per frame; graph replay {gus:.1f} us")
h, w = loop(lambda k: step(k % nview), 640)
print(f"moving camera, 64 views in turn: host {h:.1f} us, wall {w:.1f} us per frame = {1e6 / w:.0f} frames/s")
with torch.autograd.set_multithreading_enabled(False):      # (the node's backward in the calling thread: no hand-over to the device's autograd thread)
    h, w = loop(lambda k: step(k % nview), 640)
print(f"the same with torch.autograd.set_multithreading_enabled(False): host {h:.1f} us, wall {w:.1f} us per frame = {1e6 / w:.0f} frames/s")
