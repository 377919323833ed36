import numpy as np, torch, sys
sys.path.insert(0, '.')
from voge_amd import _lib, scenes, ops
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform, pixel_rays
lib = _lib.load()
dev = 'cuda:0'
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS['cfg3_50k_512']
verts, sig, cols = scenes.random_gaussians(N, seed=0)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev, R=R, T=T)
rays, origin = pixel_rays(cams, (H, W))
mus = (torch.from_numpy(verts).to(dev)[None] - origin[:, None]).reshape(-1, 3).contiguous()
isg = (2 * torch.from_numpy(sig).to(dev))[:, None, None] * torch.eye(3, device=dev)
isg = isg.contiguous()
nb = lib.voge_trace_workspace_bytes(1, N, H, W)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
out = [torch.empty((1, H, W, K), dtype=d, device=dev) for d in (torch.int32, torch.float32, torch.float32, torch.float32)]
P = lambda x: x.data_ptr()
rc = lib.voge_trace_topk_fwd(P(mus), P(isg), P(rays), None, 1, N, H, W, K, 4.6051702, P(ws), nb, *[P(o) for o in out], None, None)
torch.cuda.synchronize()
# locate bin_lb: layout cull(P*16), evr(P*48), bin_count, bin_id, bin_lb  (256-aligned)
al = lambda v: (v + 255) & ~255
nbin = ((W + 63) // 64) * ((H + 63) // 64)
ntile = ((W + 7) // 8) * ((H + 7) // 8)
off = al(N * 16) + al(N * 48) + al(nbin * 4) + 2 * al(nbin * 8192 * 4) + al(ntile * 4) + al(ntile * 2048 * 4) + 3000 * 2048 * 4
d = ws[off: off + 32].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
tb = ws[off + 256: off + 256 + 2048 * 32].view(torch.int64).cpu().numpy().reshape(2048, 4)
waves = (H // 8) * (W // 8)
print('rc', rc, 'waves', waves)
print('candidates evaluated per wave', d[0] / waves)
print('lane inserts: append', d[1], 'middle', d[2], 'full-replace', d[3], 'rejected(key>=worst)', d[4])
print('per pixel: append %.1f middle %.1f full %.1f rejected %.1f' % tuple(d[1:5] / (H * W)))
print('waves exited early', d[5], 'tile list length avg', d[6] / waves)
print('hits/pixel', (out[0] >= 0).sum().item() / (H * W))

t0 = tb[:, 0].min()
st, mid, en, n = (tb[:, 0] - t0) / 100.0, (tb[:, 1] - t0) / 100.0, (tb[:, 2] - t0) / 100.0, tb[:, 3]   # 100 MHz wall clock -> us
print('kernel span us', en.max(), ' sweep-phase dur: mean %.1f p50 %.1f p90 %.1f max %.1f' % ((mid - st).mean(), np.percentile(mid - st, 50), np.percentile(mid - st, 90), (mid - st).max()))
print('epilogue dur mean %.1f max %.1f' % ((en - mid).mean(), (en - mid).max()))
dur = mid - st
for lo, hi in ((0, 1), (1, 50), (50, 150), (150, 250), (250, 400), (400, 3000)):
    m = (n >= lo) & (n < hi)
    if m.any(): print('list len [%d,%d): %d WGs, sweep dur mean %.1f us, start mean %.1f' % (lo, hi, m.sum(), dur[m].mean(), st[m].mean()))
# concurrency over time
ts = np.linspace(0, en.max(), 20)
print('concurrent WGs at times:', [(round(t), int(((st <= t) & (en > t)).sum())) for t in ts])
