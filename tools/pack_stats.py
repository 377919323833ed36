"""Hit-count statistics of a config's frame and a cost model of lane layouts for the composite / fragment-backward
kernels: a round of a wave costs ~ its longest pixel (window walks are wave-uniform loops), so
cost(layout) = sum over rounds of max(lanes-per-pixel in the round).  usage: python tools/pack_stats.py [config]"""
import sys
import numpy as np
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
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1, max_point_per_bin=-1)).to(dev)
with torch.no_grad():
    frag = renderer(gm, R=R, T=T)
idx = frag.vert_index[0]
cnt = (idx >= 0).sum(-1).cpu().numpy().astype(np.int64)          # [H, W]
ln = frag.vert_hit_length[0]
# window trips of the walk: per pixel, the largest number of list entries within the pixel's window radius of a row
th = renderer  # (unused)
print(name, "pixels", cnt.size, "lit", (cnt > 0).mean().round(3), "mean hits (lit)", cnt[cnt > 0].mean().round(2), "full", (cnt == K).mean().round(3))
print("hist of hits/4:", np.bincount(cnt.ravel() // 4, minlength=K // 4 + 1))
np.save("gpurun_out/cnt_%s.npy" % name, cnt.astype(np.int16))


def rounds_fixed(cnt, NS, pw, cols, rows):
    """old layout: a wave = cols x rows rounds of pw pixels along x; a round is skipped when its pixels are empty"""
    LP = (K + NS - 1) // NS
    need = (cnt + NS - 1) // NS
    Hh, Ww = need.shape
    bw = pw * cols
    cost = 0; nr = 0
    for x in range(0, Ww, pw):
        m = need[:, x:x + pw].max(axis=1)
        cost += m.sum(); nr += (m > 0).sum()
    return cost, nr


def rounds_packed(cnt, NS, gw, gh, sort):
    need = (cnt + NS - 1) // NS
    Hh, Ww = need.shape
    cost = 0; nr = 0
    for y in range(0, Hh, gh):
        for x in range(0, Ww, gw):
            g = need[y:y + gh, x:x + gw].ravel()
            g = g[g > 0]
            if sort:
                g = np.sort(g)[::-1]
            used = 0; mx = 0
            for n in g:
                if used + n > 64:
                    cost += mx; nr += 1; used = 0; mx = 0
                used += n; mx = max(mx, n)
            if used:
                cost += mx; nr += 1
    return cost, nr


for NS, pw in ((2, 3), (4, 6)):
    c, r = rounds_fixed(cnt, NS, pw, 1, 1)
    print(f"NS={NS} fixed {pw} px/round: cost {c} rounds {r}")
    for gw, gh in ((4, 3), (4, 4), (8, 4), (8, 8), (16, 1), (64, 1)):
        for sort in (False, True):
            c, r = rounds_packed(cnt, NS, gw, gh, sort)
            print(f"NS={NS} packed {gw}x{gh} sort={sort}: cost {c} rounds {r}")
