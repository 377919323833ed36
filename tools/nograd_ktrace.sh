#!/bin/bash
# On the GPU box: per-kernel times of tools/cliff_nograd.py.  usage: tools/nograd_ktrace.sh N size K dist rlo rhi
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/nk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/cliff_nograd.py "$@" > gpurun_out/cliff_nograd.txt 2>&1
python tools/rocprof_summary.py $OUT gpurun_out/nograd_ktrace.txt > /dev/null
rm -rf $OUT
tail -6 gpurun_out/cliff_nograd.txt
head -10 gpurun_out/nograd_ktrace.txt | cut -c1-70,90-150
