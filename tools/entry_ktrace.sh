#!/bin/bash
# On the GPU box: kernel durations AND the gaps between the kernels of the stand-alone trace entry points.
# usage: tools/entry_ktrace.sh [stage ...]   (default: trace_fwd trace_lean_fwd)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
STAGES=${@:-trace_fwd trace_lean_fwd}
for st in $STAGES; do
  OUT=gpurun_out/ektrace_$st
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 5 --warmup 2 --only-stage $st > $OUT/bench.out 2>&1
  python3 - "$OUT" "$st" <<'PY'
import csv, glob, os, sys
d, st = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
short = lambda n: n.split("(")[0].replace("void ", "").replace("voge::", "")[:40]
# chains: binA -> binB -> sweep
import collections
dur = collections.defaultdict(list); gap = collections.defaultdict(list); chain = []
for i, r in enumerate(rows):
    n = short(r["Kernel_Name"])
    if not any(k in n for k in ("binA", "binB", "trace_fwd_kernel", "sweep_iso")):
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append((e - s) / 1e3)
    if "binA" in n:
        chain = [(n, s, e)]
    elif chain:
        gap[chain[-1][0] + " -> " + n].append((s - chain[-1][2]) / 1e3)
        chain.append((n, s, e))
        if "binB" not in n:
            tot = (e - chain[0][1]) / 1e3
            dur["chain binA start -> sweep end"].append(tot)
            chain = []
print("==", st)
for k, v in dur.items():
    v = v[len(v) // 3:]      # (skip the warm-up calls)
    print(f"  {k:46s} n={len(v):3d} avg {sum(v) / len(v):7.2f} us  min {min(v):7.2f}")
for k, v in gap.items():
    v = v[len(v) // 3:]
    print(f"  gap {k:70s} avg {sum(v) / len(v):6.2f} us")
PY
  rm -rf $OUT
done
