import sys, time, cProfile, pstats, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
dev = torch.device("cuda", 0)
import os
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
if os.environ.get("TINY"):      # TINY=1: 64 Gaussians on 16x16 pixels -- the GPU work vanishes, what is left is the host's cost per frame
    N, (H, W), focal, pp = 64, (16, 16), 20.0, (8.0, 8.0)
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
def step():
    for p in params: p.grad = None
    to_white_background(renderer(gm, R=R, T=T), colors).sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(50): step()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host per step %.3f ms, total %.3f ms"%((t1-t0)/50*1e3,(t2-t0)/50*1e3))
pr=cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
