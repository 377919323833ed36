#!/bin/bash
# On the GPU box: per-kernel times of the bench frame alone (eager launches) -> gpurun_out/frame_ktrace.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/fk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 100 --warmup 5 --only-stage frame "$@" > /dev/null 2>&1
python tools/rocprof_summary.py $OUT gpurun_out/frame_ktrace.txt > /dev/null
rm -rf $OUT
head -16 gpurun_out/frame_ktrace.txt | cut -c1-70,90-150
