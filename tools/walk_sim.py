"""The fused backward's window walks, simulated on the fragments of a config with the kernel's own lane packing
(fragment_bwd.hip: groups of 4 x 3 pixels, a lane owns two consecutive slots, rounds of as many consecutive pixels as fit 64
lanes; composite_core.h compn_bwd_wave<2, true>: a walk over the row pairs BEHIND the lane's columns, then one over the pairs
IN FRONT, each as long as the wave's longest lane).  Prints the wave iterations of
  (a) today's form:        sum over rounds of  max_lane(behind) + max_lane(front)
  (b) one merged walk:     sum over rounds of  max_lane(behind + front)        -- a lane walks its two directions back to back
  (c) a perfectly flat one: sum over rounds of ceil(sum_lane(behind + front) / 64)
and the share of lanes active in (a) -- to be compared with the kernel's own counter (profiles/r5_fragment_bwd_sections_final.txt:
310 620 wave iterations, 41 % of lanes active).   usage (GPU box): python tools/walk_sim.py [config]"""
import sys
import torch
sys.path.insert(0, ".")
from voge_amd import scenes, ops
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
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
with torch.no_grad():
    frag = renderer(gm, R=R, T=T)
    lz = frag._lazy
    act, dsd = ops._act_dsd([None, None], lz.records, lz.rays, lz.sel_idx, lz.sel_len, lz.cnt, lz.B * lz.N)
ln = lz.sel_len[0]                      # [H, W, K]
cnt = lz.cnt[0].clamp(0, K)             # [H, W]
s = torch.sqrt(dsd[0] + 1e-10)
kSat = 3.5                               # (composite_core.h: kSat / s is a column's window radius in len)
GW, GH = 4, 3
NSL = 2
LP = (K + 1) // 2
k = torch.arange(K, device=dev)
live = k[None, None, :] < cnt[..., None]
rad = torch.where(live, kSat / s, torch.zeros_like(s))                    # [H, W, K] column radius (0: an empty column needs no rows)
BIG = 1e30
lnp = torch.where(live, ln, torch.full_like(ln, BIG))                    # rows behind the list: the kernel's +big sentinel
# per lane (pixel, q): columns 2q, 2q+1.  behind trips = number of row pairs t >= 1 (pair start row 2(q+t)) with
# len[pair start] - len[col] < rad[col] for either own column; front trips likewise with the pair's SECOND row (the nearer one)
q = torch.arange(LP, device=dev)
col = (2 * q[:, None] + torch.arange(2, device=dev)[None, :]).clamp(max=K - 1)      # [LP, 2]
lcol = lnp[..., col]                                                      # [H, W, LP, 2]
rcol = torch.where((2 * q[:, None] + torch.arange(2, device=dev)[None, :]) < K, rad[..., col], torch.zeros_like(rad[..., col]))
pair_first = lnp[..., 0::2]                                              # [H, W, K/2]  len of rows 0, 2, 4, ...
pair_second = torch.where(live[..., 1::2], ln[..., 1::2], torch.full_like(ln[..., 1::2], BIG)) if K % 2 == 0 else None
npair = pair_first.shape[-1]
t = torch.arange(npair, device=dev)
# behind: pairs with index > q
need_b = ((pair_first[..., None, :, None] - lcol[..., :, None, :]) < rcol[..., :, None, :]).any(-1) & (t[None, None, None, :] > q[None, None, :, None])
behind = need_b.sum(-1)                                                   # [H, W, LP]  (windows are contiguous: the count is the trip count)
ps = ln[..., 1::2]
ps = torch.where(live[..., 1::2], ps, torch.full_like(ps, -BIG))          # (front rows are always live when the column is)
need_f = ((lcol[..., :, None, :] - ps[..., None, :, None]) < rcol[..., :, None, :]).any(-1) & (t[None, None, None, :] < q[None, None, :, None])
front = need_f.sum(-1)
lanes_need = (cnt + 1) // 2                                               # [H, W]
on = q[None, None, :] < lanes_need[..., None]
behind = behind * on
front = front * on
# groups of GW x GH pixels, rounds by prefix sums (pack_round)
Hp, Wp = (H + GH - 1) // GH * GH, (W + GW - 1) // GW * GW
def pad(x, fill=0):
    out = torch.full((Hp, Wp) + x.shape[2:], fill, dtype=x.dtype, device=dev)
    out[:H, :W] = x
    return out
need_g = pad(lanes_need).reshape(Hp // GH, GH, Wp // GW, GW).permute(0, 2, 1, 3).reshape(-1, GH * GW)
b_g = pad(behind).reshape(Hp // GH, GH, Wp // GW, GW, LP).permute(0, 2, 1, 3, 4).reshape(-1, GH * GW, LP)
f_g = pad(front).reshape(Hp // GH, GH, Wp // GW, GW, LP).permute(0, 2, 1, 3, 4).reshape(-1, GH * GW, LP)
need_c, b_c, f_c = need_g.cpu(), b_g.cpu(), f_g.cpu()
import numpy as np
need_c, b_c, f_c = need_c.numpy(), b_c.numpy(), f_c.numpy()
it_a = it_b = it_c = 0
it_cap = {1.0: 0.0, 1.5: 0.0, 2.0: 0.0, 3.0: 0.0}      # (d) capped walks + the long lanes' rest taken over by the whole wave


def capped(tr, c2):
    """min over T of (T + 1) + c2 * #{lanes with more than T trips}: the walk stops at T, every lane that is not done gets ONE
    wave-wide pass (its remaining row pairs spread over the 64 lanes, a wave reduction back to it) priced at c2 iterations."""
    tr = np.sort(tr)[::-1]
    best = tr[0] + 1.0
    for k in range(1, min(len(tr), 24)):      # the k heaviest lanes go to the second phase: T = the (k+1)-th largest
        T = tr[k] if k < len(tr) else 0
        best = min(best, T + 1.0 + c2 * k)
    return best

lane_iters = 0
rounds = 0
for g in range(need_c.shape[0]):
    nd = need_c[g]
    if nd.sum() == 0:
        continue
    p = 0
    while p < nd.shape[0]:
        used, p0 = 0, p
        while p < nd.shape[0] and used + nd[p] <= 64:
            used += nd[p]
            p += 1
        if used == 0:
            continue
        bb = np.concatenate([b_c[g, i, :nd[i]] for i in range(p0, p) if nd[i] > 0])
        ff = np.concatenate([f_c[g, i, :nd[i]] for i in range(p0, p) if nd[i] > 0])
        # (+1: the trip that finds the window's end)
        it_a += (bb.max() + 1) + (ff.max() + 1)
        it_b += (bb + ff).max() + 2
        it_c += -(-int((bb + ff + 2).sum()) // 64)
        for c2 in it_cap:
            it_cap[c2] += capped(bb, c2) + capped(ff, c2)
        lane_iters += int((bb + ff + 2).sum())
        rounds += 1
print(f"{name}: {rounds} rounds; wave iterations  today {it_a}  merged walk {it_b} ({it_b / it_a:.2f})  flat {it_c} ({it_c / it_a:.2f}); "
      f"lanes active today {lane_iters / (64.0 * it_a):.2f}, merged {lane_iters / (64.0 * it_b):.2f}")
print("capped walks, the long lanes' rest as one wave-wide pass each, priced at c2 iterations per long lane:  " +
      "  ".join(f"c2={c2}: {v:.0f} ({v / it_a:.2f})" for c2, v in it_cap.items()))
