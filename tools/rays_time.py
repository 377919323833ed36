"""In-graph duration of the ray kernel (50 launches captured into a HIP graph and replayed): with the cone hierarchy, without it,
and the stand-alone cone kernel.  usage (GPU box): python tools/rays_time.py"""
import sys, torch
sys.path.insert(0, ".")
from voge_amd import _lib
from voge_amd.cameras import look_at_view_transform
dev = torch.device("cuda", 0)
lib = _lib.load()
R, T = look_at_view_transform(dist=4.0, elev=10.0, azim=70.0, device=dev)
P = lambda t: None if t is None else t.data_ptr()
for H in (128, 512, 1024):
    f = torch.tensor([[600.0, 600.0]], device=dev); pp = torch.tensor([[H / 2, H / 2]], device=dev)
    rays = torch.empty((1, H, H, 3), device=dev); o = torch.empty((1, 3), device=dev)
    cones = torch.empty((int(lib.voge_cones_floats(1, H, H)),), device=dev)
    calls = {
        "rays + cone hierarchy": lambda st: lib.voge_rays_fwd(P(R), P(T), P(f), P(pp), 1, 0, H, H, P(rays), P(o), P(cones), st),
        "rays alone": lambda st: lib.voge_rays_fwd(P(R), P(T), P(f), P(pp), 1, 0, H, H, P(rays), P(o), None, st),
        "cones from rays (voge_ray_cones)": lambda st: lib.voge_ray_cones(P(rays), 1, H, H, P(cones), st),
    }
    for name, call in calls.items():
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            call(s.cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                assert call(torch.cuda.current_stream().cuda_stream) == 0
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{H}x{H}: {name:36s} {e0.elapsed_time(e1) * 1e3 / 1000:.2f} us per launch")
