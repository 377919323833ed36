#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (voge:: kernels only) from counter_collection.csv files."""
import csv
import glob
import sys
from collections import defaultdict


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "voge::" not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
    for k in sorted(acc):
        print(f"{k}  vgpr={meta[k][0]} lds={meta[k][1]} wg={meta[k][2]} grid={meta[k][3]}")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"    {c:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1:])
