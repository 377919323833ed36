"""What limits the fused backward's window loops: a simulation of its rounds on a real frame's fragments.
For every lit pixel the per-lane trip counts of the four window loops (composite_core.h: compn_bwd_wave, NS = 2) are
computed from (len, s); rounds are formed as the kernel forms them (4x3 pixel groups, pack_round: consecutive pixels while
their lanes fit 64), and the cost of a round = wave-level trip counts (max over its lanes) x instructions per iteration.
Orderings of a group's pixels before packing: as stored / sorted by hit count / sorted by the pixel's own widest window.
usage (GPU box): python tools/window_sim.py [config]"""
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
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
with torch.no_grad():
    frag = renderer(gm, R=R, T=T)
    idx = frag.vert_index[0].long().clamp_min(0)
    cnt = frag.valid_num[0]
    ln = frag.vert_hit_length[0]
    s = torch.sqrt(2.0 * gm.sigmas.detach())[idx]            # dsd = a |d|^2 = 2 sigma (unit rays); s = sqrt(dsd)
    live = torch.arange(K, device=dev)[None, None] < cnt[..., None]
    rad = torch.where(live, 3.5 / s, torch.zeros_like(s))     # window radius of a column, in length units
    BIG = 3.0e38
    lnz = torch.where(live, ln, torch.full_like(ln, BIG))
    # per slot: rows behind / in front inside the column's own radius; per pixel: widest radius (the row loops' window)
    d = lnz[..., None, :] - lnz[..., :, None]                # d[b, m] = len_m - len_b
    m_idx = torch.arange(K, device=dev)
    behind = ((d < rad[..., :, None]) & (m_idx[None, :] > m_idx[:, None]) & live[..., None, :] & live[..., :, None]).sum(-1)
    front = ((-d < rad[..., :, None]) & (m_idx[None, :] < m_idx[:, None]) & live[..., None, :] & live[..., :, None]).sum(-1)
    rwin = rad.max(-1).values
    rbehind = ((d < rwin[..., None, None]) & (m_idx[None, :] > m_idx[:, None]) & live[..., None, :] & live[..., :, None]).sum(-1)
    rfront = ((-d < rwin[..., None, None]) & (m_idx[None, :] < m_idx[:, None]) & live[..., None, :] & live[..., :, None]).sum(-1)
behind, front, rbehind, rfront, cnt = (x.cpu().numpy() for x in (behind, front, rbehind, rfront, cnt))
# per lane (2 slots): iterations of 2 rows each; the lane's own second slot is its diagonal block, not a loop iteration
def lane_trips(x, fwd):                                       # x [H, W, K] neighbours per slot -> [H, W, K/2] loop iterations
    a, b = x[..., 0::2], x[..., 1::2]
    if fwd:      # behind: slot 2q has (its count - 1) rows beyond the lane's own pair, slot 2q+1 has its count
        n = np.maximum(a - 1, b)
    else:        # in front: slot 2q+1 has (count - 1) beyond the own pair
        n = np.maximum(a, b - 1)
    return (np.maximum(n, 0) + 1) // 2
cb, cf, rb, rf = lane_trips(behind, True), lane_trips(front, False), lane_trips(rbehind, True), lane_trips(rfront, False)
GW, GH = 4, 3
COL, ROW = 58, 20                                             # VALU per iteration (assembly of fragment_bwd_kernel)

def simulate(order):
    tot_wave = tot_lane = 0.0
    for y0 in range(0, H, GH):
        for x0 in range(0, W, GW):
            pix = [(y, x) for y in range(y0, min(y0 + GH, H)) for x in range(x0, min(x0 + GW, W))]
            c = np.array([cnt[p] for p in pix])
            if c.sum() == 0:
                continue
            if order == "count":
                o = np.argsort(-c, kind="stable")
            elif order == "window":
                key = np.array([(cb[p] + cf[p]).max() if cnt[p] else 0 for p in pix])
                o = np.argsort(-key, kind="stable")
            else:
                o = np.arange(len(pix))
            lanes = 0
            cur = []
            rounds = []
            for i in o:
                need = (c[i] + 1) // 2
                if need == 0:
                    continue
                if lanes + need > 64:
                    rounds.append(cur); cur = []; lanes = 0
                cur.append(pix[i]); lanes += need
            if cur:
                rounds.append(cur)
            for rnd in rounds:
                mcb = max(cb[p][: (cnt[p] + 1) // 2].max() for p in rnd)
                mcf = max(cf[p][: (cnt[p] + 1) // 2].max() for p in rnd)
                mrb = max(rb[p][: (cnt[p] + 1) // 2].max() for p in rnd)
                mrf = max(rf[p][: (cnt[p] + 1) // 2].max() for p in rnd)
                tot_wave += COL * (mcb + mcf) + ROW * (mrb + mrf)
                for p in rnd:
                    n = (cnt[p] + 1) // 2
                    tot_lane += (COL * (cb[p][:n] + cf[p][:n]).sum() + ROW * (rb[p][:n] + rf[p][:n]).sum()) / 64.0
    return tot_wave, tot_lane

def simulate_tail(T, c_tail=100.0):
    """Cut every window loop after T iterations; what is left (lane-iterations beyond T, 4 pairs each) is evaluated one pair
    per lane in a cooperative second phase: ceil(pairs / 64) steps of c_tail VALU (LDS reads, unpacked evaluation, segmented
    reduction of the three column sums by key, write-back)."""
    tot = 0.0
    for y0 in range(0, H, GH):
        for x0 in range(0, W, GW):
            pix = [(y, x) for y in range(y0, min(y0 + GH, H)) for x in range(x0, min(x0 + GW, W))]
            c = np.array([cnt[p] for p in pix])
            if c.sum() == 0:
                continue
            lanes, cur, rounds = 0, [], []
            for i in range(len(pix)):
                need = (c[i] + 1) // 2
                if need == 0:
                    continue
                if lanes + need > 64:
                    rounds.append(cur); cur = []; lanes = 0
                cur.append(pix[i]); lanes += need
            if cur:
                rounds.append(cur)
            for rnd in rounds:
                for arr, cost in ((cb, COL), (cf, COL), (rb, ROW), (rf, ROW)):
                    t = np.concatenate([arr[p][: (cnt[p] + 1) // 2] for p in rnd])
                    tot += cost * min(int(t.max()), T)
                    left = int(np.maximum(t - T, 0).sum()) * 4
                    if left:
                        tot += c_tail * (cost / COL) * -(-left // 64)
    return tot


def simulate_sorted_columns(col1=29.0, row1=10.0):
    """One column per lane (no own pair), the columns of a whole 4x3 group sorted by their trip counts before they are cut
    into rounds of 64 lanes: a round then holds columns of similar windows.  Trips at one column x two rows per iteration
    (col1 VALU) / one row x two columns (row1 VALU)."""
    tot = 0.0
    f1 = (front + 1) // 2; b1 = (behind + 1) // 2; rf1 = (rfront + 1) // 2; rb1 = (rbehind + 1) // 2
    for y0 in range(0, H, GH):
        for x0 in range(0, W, GW):
            pix = [(y, x) for y in range(y0, min(y0 + GH, H)) for x in range(x0, min(x0 + GW, W))]
            if sum(cnt[p] for p in pix) == 0:
                continue
            for fa, ba, cost in ((f1, b1, col1), (rf1, rb1, row1)):
                f = np.concatenate([fa[p][: cnt[p]] for p in pix]); b = np.concatenate([ba[p][: cnt[p]] for p in pix])
                o = np.argsort(-(f + b), kind="stable")
                f, b = f[o], b[o]
                for i in range(0, len(f), 64):
                    tot += cost * (int(f[i:i + 64].max()) + int(b[i:i + 64].max()))
    return tot


print(f"columns of a group sorted by window, one per lane: {simulate_sorted_columns() / 1e6:7.2f} M window-loop VALU wave-instructions")
for T in (1, 2, 3, 4, 6, 99):
    print(f"cut after {T:2d} iterations + cooperative tail: {simulate_tail(T) / 1e6:7.2f} M window-loop VALU wave-instructions")
for order in ("stored", "count", "window"):
    w, l = simulate(order)
    print(f"{order:8s}: window-loop VALU wave-instructions {w / 1e6:7.2f} M, at full lane utilisation {l / 1e6:7.2f} M "
          f"(utilisation {100 * l / w:.1f} %)")
