#!/bin/bash
# On the GPU box: per-kernel durations of ONE rank's band (rank R of N) of the bench frame, eager launches.
# usage: tools/band_ktrace.sh [config] [N] [R]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/bktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/band_trace.py ${3:-3} ${2:-8} ${1:-cfg3_50k_512} > $OUT/out.txt 2>&1
python tools/rocprof_summary.py $OUT gpurun_out/band_ktrace_summary.txt > /dev/null
rm -rf $OUT
head -24 gpurun_out/band_ktrace_summary.txt | cut -c1-70,90-150
