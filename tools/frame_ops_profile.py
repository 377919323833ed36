import sys, torch
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
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
def step():
    for p in params: p.grad = None
    to_white_background(renderer(gm, R=R, T=T), colors).sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(4): step()
    torch.cuda.synchronize()
evs = prof.events()
# list aten ops (cpu) in order for one step, with the device kernels they launched
names = [(e.name, e.device_type, round(e.device_time_total,1) if hasattr(e,'device_time_total') else 0) for e in evs if 'Memset' in e.name or 'memset' in e.name or e.name.startswith('aten::zero') or e.name.startswith('aten::fill') or 'zeros' in e.name]
from collections import Counter
print(Counter(n for n,_,_ in names))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
