"""What a launch ORDER would buy the fused backward: list scheduling of its waves (measured durations, tools/fb_wall.py
DUMP=file) onto the slots the kernel was seen to hold, in raster order (what the kernel does), longest first by the true
duration (the bound), and longest first by estimates available before the launch (sums over a group's pixels' hit counts).
usage: python tools/fb_order_sim.py dump.npz [slots]"""
import heapq, sys
import numpy as np
z = np.load(sys.argv[1])
t, cnt, H, W, K = z["t"], z["cnt"].astype(np.int64), int(z["H"]), int(z["W"]), int(z["K"])
GW, GH = 4, 3
bx, by = (W + GW - 1) // GW, (H + GH - 1) // GH
ng = bx * by
ran = t[:ng, 0] >= 0
d = np.where(ran, t[:ng, 1] - t[:ng, 0], 0.0)
pad = np.zeros((by * GH, bx * GW), dtype=np.int64); pad[:H, :W] = np.minimum(cnt, K)
grp = pad.reshape(by, GH, bx, GW).transpose(0, 2, 1, 3).reshape(ng, GH * GW)
lanes = ((grp + 1) // 2).sum(1)                   # lanes the group's pixels need (two slots each)
rounds = np.ceil(lanes / 64.0)
sq = (grp * grp).sum(1)
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3700
print(f"{ran.sum()} working waves of {ng}; sum of durations {d.sum():.0f} wave-us; perfect packing on {slots} slots {d.sum() / slots:.1f} us; "
      f"measured span {t[:ng][ran][:, 1].max():.1f} us")
print("correlation of the duration with: lanes %.3f, rounds %.3f, sum cnt^2 %.3f" % tuple(np.corrcoef(d[ran], x[ran])[0, 1] for x in (lanes, rounds, sq)))


def makespan(order):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for g in order:
        if d[g] <= 0:
            continue
        s = heapq.heappop(h)
        e = s + d[g]
        end = max(end, e)
        heapq.heappush(h, e)
    return end


idx = np.arange(ng)
print(f"raster order                        {makespan(idx):7.1f} us")
print(f"longest first, true durations       {makespan(np.argsort(-d, kind='stable')):7.1f} us")
print(f"longest first by lanes              {makespan(np.argsort(-lanes, kind='stable')):7.1f} us")
print(f"longest first by sum cnt^2          {makespan(np.argsort(-sq, kind='stable')):7.1f} us")
for nb in (8, 16, 32):
    b = np.minimum(lanes * nb // (GW * GH * ((K + 1) // 2) + 1), nb - 1)
    print(f"{nb:2d} buckets of lanes, raster inside   {makespan(np.argsort(-b, kind='stable')):7.1f} us")
rng = np.random.default_rng(0)
print(f"random order                        {makespan(rng.permutation(ng)):7.1f} us")
