"""Host cost of one eager forward + backward frame through the public API, measured where the GPU cannot be the bound: a scene of
a few hundred Gaussians at 64^2 (the GPU needs ~60 us for it), so the loop's wall time per frame IS the host's enqueue time of
the same Python / C path the 512^2 frame takes.  Prints the per-frame time for the white-background step and for the training
pattern, then cProfile's top entries of the former.   usage (GPU box): python tools/host_frame_cost.py [frames]"""
import cProfile
import pstats
import sys
import time
import torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, interpolate_attr, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
verts, sig, cols = scenes.random_gaussians(600, seed=1, r_lo=0.08, r_hi=0.16)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
nview = 64
Rm, Tm = look_at_view_transform(dist=[3.5] * nview, elev=[15.0] * nview, azim=[40.0 + 0.5 * i for i in range(nview)], device=dev)
cams = PerspectiveCameras(focal_length=60.0, principal_point=((32.0, 32.0),), image_size=((64, 64),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(64, 64), max_assign=16, max_point_per_bin=-1)).to(dev)
params = (gm.verts, gm.sigmas, colors)


def white(i):
    for p in params:
        p.grad = None
    to_white_background(renderer(gm, R=Rm[i:i + 1], T=Tm[i:i + 1]), colors).sum().backward()


def pattern(i):
    for p in params:
        p.grad = None
    frag = renderer(gm, R=Rm[i:i + 1], T=Tm[i:i + 1])
    (interpolate_attr(frag, colors).sum() + get_silhouette(frag).sum()).backward()


for name, fn in (("to_white_background", white), ("interpolate_attr + get_silhouette", pattern)):
    for i in range(200):
        fn(i % nview)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        fn(i % nview)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {1e6 * (t1 - t0) / frames:.1f} us per frame (wall {1e6 * (t2 - t0) / frames:.1f} us: the GPU was {'not ' if (t2 - t1) < 0.05 * (t1 - t0) else ''}the bound)")
pr = cProfile.Profile()
pr.enable()
for i in range(1000):
    white(i % nview)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
