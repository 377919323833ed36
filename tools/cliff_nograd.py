"""One case of tools/cliff_scan.py, forward only under torch.no_grad() (the path an inference caller takes).
usage: python tools/cliff_nograd.py N size K dist rlo rhi"""
import sys, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
N, size, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dist, rlo, rhi = float(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
dev = torch.device("cuda", 0)
verts, sig, cols = scenes.random_gaussians(N, seed=0, r_lo=rlo, r_hi=rhi)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
R, T = look_at_view_transform(dist=dist, elev=10.0, azim=70.0, device=dev)
cams = PerspectiveCameras(focal_length=1.17 * size, principal_point=((size / 2.0, size / 2.0),), image_size=((size, size),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(size, size), max_assign=K, max_point_per_bin=-1)).to(dev)
with torch.no_grad():
    for rep in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        idx = renderer(gm, R=R, T=T).vert_index
        e1.record()
        torch.cuda.synchronize()
        print(rep, round(e0.elapsed_time(e1) * 1e3, 1), "us")
