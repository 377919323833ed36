"""Time of the backward half of the bench frame (HIP graph of fwd+bwd minus a graph of fwd only) and the frame rate.
usage: VOGE_HIP_LIB=build/variants/x.so python tools/fb_time.py [config]"""
import sys, time, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
def fwd(): return to_white_background(renderer(gm, R=R, T=T), colors)
def both():
    for p in params: p.grad = None
    fwd().sum().backward()
def graph_time(fn, n=50):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
tb = graph_time(both)
with torch.no_grad():
    tf = graph_time(lambda: fwd().sum())
print(f"frame {tb:.1f} us ({1e6 / tb:.0f} fps)  forward {tf:.1f} us  backward {tb - tf:.1f} us")
