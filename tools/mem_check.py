import sys, torch, gc
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background, get_silhouette
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
dev = torch.device("cuda", 0)
ma = lambda: round(torch.cuda.memory_allocated() / 1e6, 1)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg5_shapefit_128"]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
r = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
print("setup", ma())
frag = r(gm, R=R, T=T); torch.cuda.synchronize(); print("after renderer", ma())
img = to_white_background(frag, colors); print("after shade", ma())
loss = img.sum() + get_silhouette(frag).sum(); loss.backward(); torch.cuda.synchronize(); print("after backward", ma())
del frag, img, loss; gc.collect(); print("after del", ma())
from voge_amd import _lib
print("workspace bytes", _lib.load().voge_trace_workspace_bytes(1, N, H, W) / 1e6, "bwd ws", _lib.load().voge_fragment_bwd_workspace_bytes(N) / 1e6)
