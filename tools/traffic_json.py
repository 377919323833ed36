#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately,
as MI355X_MICROARCH.md's HBM section prescribes).
usage: traffic_json.py <fetch_dir> <write_dir> <out.json> <config name> [frame]   (other entries of out.json are kept;
"frame": the passes ran the whole bench frame -> stored as <config>["frame_kernels"], the trace entry's total untouched)
Units: counters are KiB; FETCH_SIZE x2 on gfx950 for wide coalesced reads (same guide); per launch averages."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

TRACE_FWD = ("cones_kernel", "prep_kernel", "binA_kernel", "binB_kernel", "trace_fwd_kernel", "sweep_iso_kernel")


def collect(d, counter, frame_only=False):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "voge::" not in k or r["Counter_Name"] != counter:
                continue
            k = k.split("(")[0].replace("void ", "").replace("voge::", "")
            acc[k].append(float(r["Counter_Value"]))
    if frame_only and acc:
        # the frame's kernels run once per step; what bench.py's set-up launched once or twice (the stand-alone composite /
        # shade kernels behind its reference tensors) is not part of the frame
        most = max(len(v) for v in acc.values())
        acc = {k: v for k, v in acc.items() if 2 * len(v) >= most}
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(fetch_dir, write_dir, out, config, what="trace"):
    fe, wr = collect(fetch_dir, "FETCH_SIZE", what == "frame"), collect(write_dir, "WRITE_SIZE", what == "frame")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f, w = fe.get(k, 0.0), wr.get(k, 0.0)
        kernels[k] = {"fetch_KiB_raw": round(f), "write_KiB": round(w), "hbm_bytes": int((2 * f + w) * 1024)}
    total = sum(v["hbm_bytes"] for k, v in kernels.items() if k.split("<")[0] in TRACE_FWD)
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --no-graph. Units KiB per launch; "
                   "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction of MI355X_MICROARCH.md's HBM section).")
    if what == "frame":
        doc.setdefault(config, {})["frame_kernels"] = kernels
        doc[config]["frame_hbm_bytes"] = sum(v["hbm_bytes"] for v in kernels.values())
        print(config, "frame_hbm_bytes", doc[config]["frame_hbm_bytes"])
    else:
        doc[config] = {"kernels": kernels, "voge_trace_topk_fwd_kernels": list(TRACE_FWD), "voge_trace_topk_fwd_bytes": total}
        print(config, "voge_trace_topk_fwd_bytes", total)
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:6])
