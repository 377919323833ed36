import os, sys, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[os.environ.get("CFG", "cfg5_shapefit_128")]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
rows = (40, 90)
def fwd_only():
    return to_white_background(renderer(gm, R=R, T=T, rows=rows), colors)
import os
if os.environ.get("REF"):
    ref_b = fwd_only(); ref_g = torch.autograd.grad(ref_b, params, torch.ones_like(ref_b))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        b = fwd_only(); torch.autograd.grad(b, params, torch.ones_like(b))
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    bs = fwd_only()
ones = torch.ones_like(bs)
with torch.cuda.graph(g2, pool=g1.pool()):
    gs = torch.autograd.grad(bs, params, ones)
for _ in range(3):
    g1.replay(); g2.replay()
torch.cuda.synchronize()
print("replays ok"); ref_b = fwd_only(); ref_g = torch.autograd.grad(ref_b, params, torch.ones_like(ref_b)); print("band equal", torch.equal(bs, ref_b))
for a, b in zip(gs, ref_g):
    print("grad max abs diff", (a - b).abs().max().item(), "scale", b.abs().max().item())
