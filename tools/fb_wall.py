"""When the fused backward's waves start and end (debug build -DVOGE_FB_TIMES: first / last s_memrealtime stamp per wave):
is the kernel as long as its work, or as long as a tail of late heavy waves?
usage on the GPU box: VOGE_HIP_LIB=build/variants/fbt.so python tools/fb_wall.py [config]"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from voge_amd import _lib, scenes
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from voge_amd.Renderer import GaussianRenderSettings, GaussianRenderer, to_white_background
from voge_amd.Meshes import GaussianMeshes
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, colors = scenes.random_gaussians(N, seed=0)
dev = torch.device("cuda", 0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
cols = torch.from_numpy(colors).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
for _ in range(3):
    to_white_background(renderer(gm, R=R, T=T), cols).sum().backward()
torch.cuda.synchronize()
nw = min(((W + 3) // 4) * ((H + 2) // 3), 1 << 16)
buf = (ctypes.c_ulonglong * (2 * nw))()
raw.voge_debug_fb_wall(buf, nw, 1)
loss = to_white_background(renderer(gm, R=R, T=T), cols).sum()
torch.cuda.synchronize()
loss.backward()
torch.cuda.synchronize()
raw.voge_debug_fb_wall(buf, nw, 0)
t = np.array(list(buf), dtype=np.float64).reshape(nw, 2) * 0.01      # us
import os
if os.environ.get("DUMP"):      # per-group stamps + the hit counts: the input of tools/fb_order_sim.py
    with torch.no_grad():
        cnt = renderer(gm, R=R, T=T).valid_num.reshape(H, W).to(torch.int16).cpu().numpy()
    np.savez_compressed(os.environ["DUMP"], t=np.where(t > 0, t - t[t[:, 0] > 0, 0].min(), -1.0).astype(np.float32), cnt=cnt, H=H, W=W, K=K)
ran = t[:, 0] > 0
t = t[ran]
t0 = t[:, 0].min()
s, e = t[:, 0] - t0, t[:, 1] - t0
d = e - s
print(f"{name}: {ran.sum()} of {nw} recorded waves ran; span {e.max():.1f} us; wave duration mean {d.mean():.2f} max {d.max():.2f} us; "
      f"sum of durations / span = {d.sum() / e.max():.0f} waves in flight on average")
edges = np.linspace(0, e.max(), 13)
print("  in flight at t (us):", [(round(float(x), 1), int(((s <= x) & (e > x)).sum())) for x in edges[:-1]])
print("  duration percentiles (us): ", {p: round(float(np.percentile(d, p)), 1) for p in (10, 50, 90, 99, 100)})
late = np.argsort(-e)[:8]
print("  last to end: ", [(round(float(s[i]), 1), round(float(d[i]), 1)) for i in late], "(start, duration)")
