"""Frame time of the GENERAL (anisotropic, [N,3,3] sigmas) path at the cfg3 size, for comparison with
bench.py's isotropic headline config."""
import sys, time, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
for aniso in (False, True):
    verts, sig, cols = scenes.random_gaussians(N, seed=0, anisotropic=aniso)
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
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("anisotropic" if aniso else "isotropic  ", f"{dt*1e3:.3f} ms/frame eager  ({1/dt:.0f} frames/s)")
