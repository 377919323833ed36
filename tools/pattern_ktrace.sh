#!/bin/bash
# On the GPU box: per-kernel times of the training pattern's frame (tools/pattern_step.py) -> gpurun_out/pattern_ktrace.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/pattern_step.py 100 > /dev/null 2>&1
python tools/rocprof_summary.py $OUT gpurun_out/pattern_ktrace.txt > /dev/null
rm -rf $OUT
head -24 gpurun_out/pattern_ktrace.txt | cut -c1-70,90-150
