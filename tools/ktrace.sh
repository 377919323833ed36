#!/bin/bash
# On the GPU box: per-kernel durations of the bench frame (eager launches) -> gpurun_out/ktrace_summary.txt
# usage: tools/ktrace.sh [extra bench.py args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 30 --warmup 5 "$@" > $OUT/bench.out 2>&1
python tools/rocprof_summary.py $OUT gpurun_out/ktrace_summary.txt > /dev/null
rm -rf $OUT
grep -E "voge" gpurun_out/ktrace_summary.txt | cut -c1-70,90-150
