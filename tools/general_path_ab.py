"""The GENERAL ([N,3,3] Sigma^-1) path against the scalar-sigma path on the SAME scene, and on bench.py's anisotropic
scene: frame time (HIP graph replay, forward + backward) and hits kept per pixel.
  case iso      : cfg3's scalar sigmas                       (the headline path)
  case iso3x3   : the same Gaussians handed over as s * I    (general kernels, identical hits)
  case aniso    : scenes.random_gaussians(anisotropic=True)  (bench.py's anisotropic_3x3 variant)
usage: python tools/general_path_ab.py case [config]      (under rocprofv3 --kernel-trace for the per-kernel split)"""
import sys, time, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
case = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "cfg3_50k_512"
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, cols = scenes.random_gaussians(N, seed=0, anisotropic=(case == "aniso"))
sig = torch.from_numpy(sig)
if case == "iso3x3":
    sig = sig[:, None, None] * torch.eye(3)[None]
gm = GaussianMeshes(torch.from_numpy(verts), sig.contiguous()).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
with torch.no_grad():
    frag = renderer(gm, R=R, T=T)
    vn = frag.valid_num.float()
    print(case, f"hits kept per pixel: mean {vn.mean().item():.1f}, full (= K) on {(vn == K).float().mean().item() * 100:.0f} % of the pixels")
def step():
    for p in params: p.grad = None
    to_white_background(renderer(gm, R=R, T=T), colors).sum().backward()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): g.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(case, f"{dt * 1e6:.0f} us per frame (graph replay) = {1 / dt:.0f} frames/s")
