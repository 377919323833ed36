#!/bin/bash
# On the GPU box: per-kernel durations of a demo script (eager launches) -> gpurun_out/demo_ktrace_<name>.txt
# usage: tools/demo_ktrace.sh <demo script> [its args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
NAME=$(basename "$1" .py)
OUT=gpurun_out/dktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/out.txt 2>&1
tail -2 $OUT/out.txt
python tools/rocprof_summary.py $OUT gpurun_out/demo_ktrace_$NAME.txt > /dev/null
rm -rf $OUT
head -28 gpurun_out/demo_ktrace_$NAME.txt | cut -c1-80,90-150
