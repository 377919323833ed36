import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from voge_amd import ops
from oracle import camera_np
R, T = camera_np.look_at_view_transform([3.0, 4.0], [10.0, -20.0], [30.0, 100.0])
R = R.astype(np.float32); T = T.astype(np.float32)
f = np.float32([[300.0, 310.0], [150.0, 140.0]]); pp = np.float32([[26.0, 18.0], [30.0, 20.5]])
H, W = 37, 53
ref, ro = camera_np.pixel_rays(R, T, f, pp, (H, W))
tt = lambda x: torch.tensor(x, device="cuda")
rays, origin = ops.pixel_rays(tt(R), tt(T), tt(f), tt(pp), 0, H, W)
d = np.abs(rays.cpu().numpy() - ref)
print(d.max(), np.argwhere(d > 1e-3)[:10], (d > 1e-3).sum())
print(rays.cpu().numpy().reshape(2, -1, 3)[1, :3], ref.reshape(2, -1, 3)[1, :3])
