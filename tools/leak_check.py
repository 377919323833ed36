import sys, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background, get_silhouette
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg5_shapefit_128"]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
r = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
def step(i):
    for p in (gm.verts, gm.sigmas, colors): p.grad = None
    frag = r(gm, R=R, T=T)
    img = to_white_background(frag, colors)
    loss = img.sum() + (get_silhouette(frag).sum() if i % 2 else 0)
    loss.backward()
for i in range(50): step(i)
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
for i in range(3000): step(i)
torch.cuda.synchronize(); m1 = torch.cuda.memory_allocated(); r1 = torch.cuda.memory_reserved()
print("allocated MB", m0 / 1e6, "->", m1 / 1e6, " reserved MB", r0 / 1e6, "->", r1 / 1e6)
