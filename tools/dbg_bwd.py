import numpy as np, torch, sys
sys.path.insert(0, '.')
from voge_amd import _lib, scenes, ops
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform, pixel_rays
lib = _lib.load(); dev = 'cuda:0'
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS['cfg3_50k_512']
verts, sig, cols = scenes.random_gaussians(N, seed=0)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev, R=R, T=T)
rays, origin = pixel_rays(cams, (H, W))
mus = (torch.from_numpy(verts).to(dev)[None] - origin[:, None]).reshape(-1, 3).contiguous()
isg = ((2 * torch.from_numpy(sig).to(dev))[:, None, None] * torch.eye(3, device=dev)).contiguous()
sel = ops.ray_trace_fine(mus, isg, rays, None, 4.6051702, 16, K)
w, vn = ops.composite(sel[0], sel[2], sel[1], sel[3], 1.0)
P = lambda x: x.data_ptr()
nb = lib.voge_trace_bwd_workspace_bytes(N)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
g_mu, g_A = torch.empty_like(mus), torch.empty_like(isg)
for _ in range(2):
    rc = lib.voge_trace_bwd(P(mus), P(isg), P(rays), P(sel[0]), None, P(w), P(w), P(w), N, H, W, K, P(ws), nb, None, P(g_mu), P(g_A), None)
torch.cuda.synchronize()
acc = ws[:N * 64].view(torch.int64).cpu().numpy().reshape(N, 8)
tb = acc[:4096, 6:8].astype(np.uint64)
t0 = (tb[:, 0] >> np.uint64(20)).astype(np.int64); dmain = (tb[:, 0] & np.uint64(0xfffff)).astype(np.int64); t2 = (tb[:, 1] & np.uint64((1 << 44) - 1)).astype(np.int64)
t0 = t0 & ((1 << 44) - 1)
base = t0.min()
st = (t0 - base) / 100.0; en = (t2 - base) / 100.0; dm = dmain / 100.0
print('span us', en.max(), 'wave total dur mean %.1f p50 %.1f p90 %.1f max %.1f' % ((en - st).mean(), np.percentile(en - st, 50), np.percentile(en - st, 90), (en - st).max()))
print('main-loop dur mean %.1f max %.1f ; flush dur mean %.1f max %.1f' % (dm.mean(), dm.max(), (en - st - dm).mean(), (en - st - dm).max()))
nvalid = (sel[0] >= 0).view(H // 8, 8, W // 8, 8, K).sum(dim=(1, 3, 4)).flatten().cpu().numpy()
for lo, hi in ((0, 1), (1, 500), (500, 1500), (1500, 2200), (2200, 2561)):
    m = (nvalid >= lo) & (nvalid < hi)
    if m.any(): print('valid slots [%d,%d): %d waves, dur mean %.1f' % (lo, hi, m.sum(), (en - st)[m].mean()))
ts = np.linspace(0, en.max(), 16)
print('concurrent waves:', [(round(t), int(((st <= t) & (en > t)).sum())) for t in ts])
