"""Measured load balancing of the pixel-row bands, tried on ONE GPU: time every band of an n-way split (forward+backward
as a HIP graph, no collectives), move the boundaries with distributed.rebalance_bounds, repeat.
usage: python tools/band_balance.py [config] [n] [rounds]"""
import sys, time, torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from voge_amd.distributed import rebalance_bounds, row_band
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]


def band_us(rows, reps=30):
    def step():
        for p in params: p.grad = None
        to_white_background(renderer(gm, R=R, T=T, rows=rows), colors).sum().backward()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): step()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


whole = band_us((0, H))
bounds = [row_band(H, r, n)[0] for r in range(n)] + [H]
fixed = None
for it in range(rounds + 1):
    times = [band_us((bounds[r], bounds[r + 1])) for r in range(n)]
    print(f"round {it}: bounds {bounds}  band us {[round(t) for t in times]}  slowest {max(times):.0f}  mean {sum(times) / n:.0f}"
          f"  compute-only efficiency {whole / max(times) / n:.2f} (whole frame {whole:.0f} us)")
    if it == rounds:
        break
    if fixed is None:
        fixed = 0.5 * min(times)          # what a nearly empty band still costs: latency, not work
    bounds = rebalance_bounds(bounds, times, fixed=fixed, damping=0.8, min_rows=8)
