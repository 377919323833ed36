"""Host time of one eager forward + backward frame (cfg3), pipelined (no drain between frames), split by the package's own
functions: perf_counter around each wrapped call, nested calls subtracted from their parent (exclusive times).
usage (GPU box): python tools/host_breakdown.py"""
import sys, time, torch
sys.path.insert(0, ".")
from voge_amd import scenes, ops, cameras, _lib
import voge_amd.Renderer as Rm
from voge_amd.Meshes import GaussianMeshes
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform

acc, stack = {}, []


def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or name

    def inner(*a, **k):
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            child = stack.pop()
            acc[label] = acc.get(label, 0.0) + dt - child
            if stack:
                stack[-1] += dt
    setattr(mod, name, inner)


def wrap_static(cls, name, label):
    fn = getattr(cls, name)

    def inner(*a, **k):
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            child = stack.pop()
            acc[label] = acc.get(label, 0.0) + dt - child
            if stack:
                stack[-1] += dt
    setattr(cls, name, staticmethod(inner))


lib = _lib.load()
for nm in ("voge_rays_striped_fwd", "voge_fragments_fwd_iso_view", "voge_composite_shade_fwd_iso", "voge_fragment_shade_bwd_iso",
           "voge_trace_workspace_bytes", "voge_fragment_bwd_workspace_bytes", "voge_cones_floats", "voge_frame_trace_fwd_iso",
           "voge_frame_shade_fwd_iso", "voge_frame_shade_bwd_iso"):
    wrap(lib, nm, "C: " + nm)
wrap(Rm, "camera_tensors", "cameras.camera_tensors (glue)")
wrap(ops, "frame_trace", "ops.frame_trace (glue)")
wrap(ops, "frame_eligible")
wrap(ops, "_tag_index")
wrap(cameras, "_intrinsics")
wrap(ops, "pixel_rays", "ops.pixel_rays (glue)")
wrap(Rm, "pixel_rays", "cameras.pixel_rays (glue)")
wrap(ops, "trace_lean", "ops.trace_lean (glue)")
wrap(ops, "composite_shade", "ops.composite_shade (glue)")
wrap(ops, "lazy_eligible")
wrap(ops, "_workspace")
wrap(ops, "check_index_range")
wrap_static(ops._PixelRays, "forward", "_PixelRays.forward (python)")
wrap_static(ops._TraceLean, "forward", "_TraceLean.forward (python)")
wrap_static(ops._CompositeShade, "forward", "_CompositeShade.forward (python)")
wrap_static(ops._CompositeShade, "backward", "_CompositeShade.backward (python)")
wrap_static(ops._TraceLean, "backward", "_TraceLean.backward (python)")

dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = Rm.GaussianRenderer(cams, Rm.GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
params = [gm.verts, gm.sigmas, colors]
tot = {"zero": 0.0, "renderer()": 0.0, "to_white_background()": 0.0, "sum()": 0.0, "backward()": 0.0}
n = 300
for it in range(n + 60):
    if it == 60:
        acc.clear()
        for k in tot: tot[k] = 0.0
        torch.cuda.synchronize(); t_all = time.perf_counter()
    t = [time.perf_counter()]
    for p in params: p.grad = None
    t.append(time.perf_counter())
    frag = renderer(gm, R=R, T=T); t.append(time.perf_counter())
    img = Rm.to_white_background(frag, colors); t.append(time.perf_counter())
    loss = img.sum(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    for k, a, b in zip(tot, t[:-1], t[1:]):
        tot[k] += b - a
t_host = time.perf_counter() - t_all
torch.cuda.synchronize()
t_wall = time.perf_counter() - t_all
print(f"per frame: host {1e6 * t_host / n:.1f} us, wall {1e6 * t_wall / n:.1f} us")
for k, v in tot.items():
    print(f"  {k:28s} {1e6 * v / n:7.1f} us")
print("exclusive times of the wrapped functions (the wrappers themselves cost ~0.5 us each):")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:46s} {1e6 * v / n:7.1f} us")
