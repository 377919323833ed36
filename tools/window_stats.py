"""How far the composite's window walks go at a config: per lit pixel, list entries inside the pixel-wide window
(3.5 / s_min, what the row walks use), inside each column's own window (3.5 / s_j, the column walks), and the pairs with
|len_m - len_j| s_j < 3.5 at all (the evaluations that matter).  usage: python tools/window_stats.py [config]"""
import sys
import torch
sys.path.insert(0, ".")
from voge_amd import scenes
from voge_amd.Meshes import GaussianMeshes
from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
dev = torch.device("cuda", 0)
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
gm.verts.requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
frag = renderer(gm, R=R, T=T)
from voge_amd import ops
th = frag.vert_weight.voge_through
ops._act_dsd(th)                      # (scalar sigmas: act / dsd are not kept by the forward; derived on request)
ln, dsd, act, idx = (th[k].reshape(-1, K) for k in ("len", "dsd", "act", "idx"))
live = idx >= 0
n = live.sum(-1)
lit = n > 0
ln, dsd, live, n, act = ln[lit], dsd[lit], live[lit], n[lit], act[lit]
s = torch.sqrt(dsd + 1e-10)
rad = torch.where(live, 3.5 / s, torch.zeros_like(s))          # column radius
rwin = rad.max(-1, keepdim=True).values
print(name, "lit pixels", int(lit.sum()), "mean hits", float(n.float().mean()))
tot_pix = tot_col = tot_true = 0
CH = 16384
for i in range(0, ln.shape[0], CH):
    l, r, rw, lv = ln[i:i + CH], rad[i:i + CH], rwin[i:i + CH], live[i:i + CH]
    d = (l[:, :, None] - l[:, None, :]).abs()                   # [p, m, j]
    pair = lv[:, :, None] & lv[:, None, :]
    tot_pix += int(((d < rw[:, :, None]) & pair).sum())         # row walks: the pixel-wide window
    tot_col += int(((d < r[:, None, :]) & pair).sum())          # column walks / exact need: column j's own window
full = int((n.long() ** 2).sum())
print("pairs per lit pixel: all %.1f  pixel-window %.1f  column-window %.1f" % (full / ln.shape[0], tot_pix / ln.shape[0], tot_col / ln.shape[0]))
w = frag.vert_weight.reshape(-1, K)[lit]
print("weight mass: slots with w > 1e-6: %.1f per lit pixel, > 1e-4: %.1f" % (float((w > 1e-6).sum(-1).float().mean()), float((w > 1e-4).sum(-1).float().mean())))
E = torch.exp(-act) * live
print("slots with E > 1e-3: %.1f" % float((E > 1e-3).sum(-1).float().mean()))
