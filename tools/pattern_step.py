"""The reference's training pattern at cfg3 (interpolate_attr + get_silhouette, one .sum() over their concatenation, backward --
bench.py's `interpolate_attr_and_silhouette` variant) launched eagerly N times: the program behind tools/pattern_ktrace.sh.
usage: python tools/pattern_step.py [steps]"""
import sys
import torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, interpolate_attr
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
for it in range(steps):
    for p in (gm.verts, gm.sigmas, colors):
        p.grad = None
    frag = renderer(gm, R=R, T=T)
    torch.cat((interpolate_attr(frag, colors), get_silhouette(frag).unsqueeze(-1)), dim=-1).sum().backward()
torch.cuda.synchronize()
print("done", steps)
