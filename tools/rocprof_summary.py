#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` directory into a short
per-kernel table (calls, total/avg/min/max duration) for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, out=None):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    lines = [f"# source: {d} ({len(files)} kernel_trace csv)", f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}"]
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{name[:90]:90s} {len(v):6d} {sum(v):12.1f} {sum(v) / len(v):10.2f} {min(v):10.2f} {max(v):10.2f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
