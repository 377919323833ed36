#!/bin/bash
# On the GPU box: per-kernel times of the batched ShapeFitting iteration (eager launches) -> gpurun_out/loop_ktrace.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/lk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 demo/ShapeFitting.py --iters 300 --rgb-on 0 > /dev/null 2>&1
python tools/rocprof_summary.py $OUT gpurun_out/loop_ktrace.txt > /dev/null
rm -rf $OUT
head -40 gpurun_out/loop_ktrace.txt | cut -c1-75,90-150
