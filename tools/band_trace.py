"""One band of the row sharding, launched eagerly (for rocprofv3 --kernel-trace).  usage: [STRIPE=32] band_trace.py r n [config]"""
import os, sys, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from voge_amd.distributed import Stripes, row_band, stripe_height
r, n = int(sys.argv[1]), int(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "cfg3_50k_512"
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
rows = row_band(H, r, n)
if int(os.environ.get('STRIPE', '0')) > 0:
    rows = Stripes(H, r, n, stripe_height(H, n, int(os.environ['STRIPE'])))
for _ in range(30):
    for p in params: p.grad = None
    to_white_background(renderer(gm, R=R, T=T, rows=rows), colors).sum().backward()
torch.cuda.synchronize()
