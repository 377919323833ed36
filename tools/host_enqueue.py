"""Host cost of ENQUEUEING one eager forward + backward frame (cfg3): the GPU is drained before every frame, the clock
stops when the last launch has been issued.  If this is below the frame's GPU time the eager frame is GPU-bound.
usage (GPU box): python tools/host_enqueue.py"""
import sys, time, torch
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
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
acc = {"zero": 0.0, "renderer": 0.0, "to_white_background": 0.0, "sum": 0.0, "backward": 0.0}
for it in range(260):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    for p in params: p.grad = None
    t.append(time.perf_counter())
    frag = renderer(gm, R=R, T=T); t.append(time.perf_counter())
    img = to_white_background(frag, colors); t.append(time.perf_counter())
    loss = img.sum(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    if it >= 60:
        for k, a, b in zip(acc, t[:-1], t[1:]):
            acc[k] += b - a
n = 200
for k, v in acc.items():
    print(f"{k:22s} {1e6 * v / n:7.1f} us")
print(f"{'enqueue, whole frame':22s} {1e6 * sum(acc.values()) / n:7.1f} us")
