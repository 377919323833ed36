"""Where does skipping binA (trace_fwd.hip: small_set_marks) stop paying?  The renderer-form trace (no act / dsd) replayed from a graph
for a range of set and image sizes.  usage (GPU box): VOGE_HIP_LIB=build/variants/<lib>.so python tools/small_set_scan.py"""
import math, os, sys, torch
sys.path.insert(0, ".")
from voge_amd import _lib, scenes
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform, pixel_rays
from voge_amd import ops
dev = torch.device("cuda", 0)
lib = _lib.load()
P = lambda t: None if t is None else t.data_ptr()
thr_act = -math.log(0.01 + 1e-10)
K = 25
R, T = look_at_view_transform(dist=2.7, elev=10.0, azim=40.0, device=dev)
for H in (128, 256, 512):
    cams = PerspectiveCameras(focal_length=126.0 * H / 128, principal_point=((H / 2, H / 2),), image_size=((H, H),), device=dev, R=R, T=T)
    with torch.no_grad():
        rays, origin = pixel_rays(cams, (H, H))
    cones = ops.cones_of(rays, 1, H, H)
    for N in (500, 1500, 2562, 4096, 6000, 8192, 12000, 16384):
        verts, sig, _ = scenes.random_gaussians(N, seed=N, r_lo=0.03, r_hi=0.08)
        mus = (torch.from_numpy(verts).to(dev) - origin[0]).contiguous()
        a = (2 * torch.from_numpy(sig).to(dev)).contiguous()
        nws = lib.voge_trace_workspace_bytes(1, N, H, H)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        idx = torch.empty((1, H, H, K), dtype=torch.int32, device=dev); ln = torch.empty((1, H, H, K), device=dev)
        cnt = torch.empty((1, H, H), dtype=torch.int32, device=dev); rec = torch.empty((N, 4), device=dev)
        call = lambda st: lib.voge_fragments_fwd_iso(P(mus), P(a), P(rays), None, P(cones), 1, N, H, H, K, thr_act, 1.0, P(ws), nws, P(idx), P(ln),
                                                     None, None, P(cnt), None, None, P(rec), st)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            assert call(s.cuda_stream) == 0
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                call(torch.cuda.current_stream().cuda_stream)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{os.path.basename(os.environ.get('VOGE_HIP_LIB', 'in-tree')):12s} {H}x{H} N={N:6d}: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per trace", flush=True)
