#!/usr/bin/env python3
"""The frame's timeline out of a rocprofv3 --kernel-trace csv directory: the kernels in start order, grouped into steps (a step
starts at binA), and for every position of the step the kernel's mean duration and the mean GAP between the previous kernel's end
and this one's start.  usage: tools/frame_gaps.py <dir> [label]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps, cur = [], None
for s, e, n in rows:
    if "binA_kernel" in n:
        cur = []
        steps.append(cur)
    if cur is not None:
        cur.append((s, e, n))
from collections import Counter
mode = Counter(map(len, steps)).most_common(1)[0][0]
steps = [st for k, st in enumerate(steps) if len(st) == mode and k and len(steps[k - 1]) == mode][5:-2]      # (whole steps of the steady state, behind a whole step)
print(f"# {sys.argv[2] if len(sys.argv) > 2 else d}: {len(steps)} steps of {len(steps[0])} launches")
n = len(steps[0])
tot_d = tot_g = 0.0
for i in range(n):
    dur = sum(st[i][1] - st[i][0] for st in steps) / len(steps) / 1e3
    gaps = [st[i][0] - st[i - 1][1] for st in steps] if i else [st[0][0] - pv[-1][1] for pv, st in zip(steps, steps[1:]) if st[0][0] - pv[-1][1] < 50000]
    gap = sum(gaps) / max(len(gaps), 1) / 1e3
    tot_d += dur; tot_g += gap
    print(f"  {steps[0][i][2].split('(')[0].replace('void ', '')[:70]:70s} duration {dur:7.2f} us   gap in front {gap:6.2f} us")
per = sorted(b[0][0] - a[0][0] for a, b in zip(steps, steps[1:]))
period = per[len(per) // 2] / 1e3
print(f"  sum of durations {tot_d:.1f} us + sum of gaps {tot_g:.1f} us = {tot_d + tot_g:.1f} us; step period {period:.1f} us")
