"""What would per-quadrant candidate streams save?  (CPU simulation on the cfg3 scene, numpy.)  Today an 8x8-px tile's wave
evaluates one candidate per step for all 64 rays; with one stream per 4x4-px quadrant (a DPP row of 16 lanes each) a step
would evaluate a different candidate per quadrant, each taken from the quadrant's own (tighter) list, and the wave would
run as many steps as its LONGEST quadrant needs.  The simulation walks every tile's depth-ordered candidate list with the
kernel's exit rule (all rays of the group hold K hits and the next bound is deeper) and counts steps both ways.
usage: python tools/quadrant_sim.py [tiles to sample]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from voge_amd import scenes
from oracle import camera_np

N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS["cfg3_50k_512"]
verts, sig, _ = scenes.random_gaussians(N, seed=0)
R, T = camera_np.look_at_view_transform(dd, el, az)
rays, origin = camera_np.pixel_rays(R, T, focal, pp, (H, W))
rays = rays[0].astype(np.float64)
mu = (verts - origin[0]).astype(np.float64)
a = 2.0 * sig.astype(np.float64)
thr = -np.log(0.01 + 1e-10)
reach = np.sqrt(thr / a)
nm = np.linalg.norm(mu, axis=1)
rng = np.random.default_rng(0)
ntile = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tiles = [(ty, tx) for ty in range(H // 8) for tx in range(W // 8)]
rng.shuffle(tiles)


def cone(d):      # bounding cone (axis, half angle) of unit directions d [n,3]
    ax = d.sum(0); ax /= np.linalg.norm(ax)
    return ax, np.arccos(np.clip((d @ ax).min(), -1, 1))


def members(ax, half):      # Gaussians whose sphere can touch a line of the cone (the kernels' conservative test, in angles)
    ang = np.arccos(np.clip((mu @ ax) / nm, -1, 1))
    return np.nonzero((ang <= half + np.arcsin(np.clip(reach / nm, 0, 1))) & (mu @ ax > 0))[0]


def walk(cand, d):      # steps until the exit rule fires for the rays d [n,3] over candidates `cand` in depth order
    order = cand[np.argsort(nm[cand])]
    t = d @ mu[order].T                                  # [n, C] len
    v2 = (nm[order] ** 2)[None] - t * t
    hit = a[order][None] * v2 < thr
    cnt = np.cumsum(hit, axis=1)                         # hits held after each candidate
    lb = nm[order] - 1.13 * reach[order].max()          # the list's monotone bound (bucket edge - 1.13 Rmax)
    # K-th len per ray after c candidates: approximate by the running K-th smallest hit len
    steps = len(order)
    full_at = np.where((cnt >= K).all(0))[0]
    if len(full_at):
        c0 = full_at[0]
        kth = np.sort(np.where(hit[:, :c0 + 1], t[:, :c0 + 1], np.inf), axis=1)[:, K - 1].max()
        later = np.nonzero(lb[c0 + 1:] > kth)[0]
        if len(later):
            steps = c0 + 1 + later[0]
    return steps, int(hit[:, :steps].sum())


tot8 = totq = totq_sum = hits8 = 0
used = 0
for ty, tx in tiles:
    if used >= ntile:
        break
    d = rays[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8].reshape(-1, 3)
    ax, half = cone(d)
    cand = members(ax, half)
    if len(cand) == 0:
        continue
    used += 1
    s8, h8 = walk(cand, d)
    tot8 += s8; hits8 += h8
    sq = []
    for qy in range(2):
        for qx in range(2):
            dq = rays[ty * 8 + 4 * qy:ty * 8 + 4 * qy + 4, tx * 8 + 4 * qx:tx * 8 + 4 * qx + 4].reshape(-1, 3)
            axq, hq = cone(dq)
            cq = np.intersect1d(cand, members(axq, hq))
            sq.append(walk(cq, dq)[0] if len(cq) else 0)
    totq += max(sq); totq_sum += sum(sq)
print(f"{used} lit tiles: steps per tile today {tot8 / used:.1f} (lane efficiency {hits8 / (64.0 * tot8):.2f}); with one stream per quadrant "
      f"{totq / used:.1f} (longest quadrant; mean quadrant {totq_sum / used / 4:.1f}) -> {100 * (1 - totq / tot8):.0f} % fewer steps")
