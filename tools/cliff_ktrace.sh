#!/bin/bash
# On the GPU box: per-kernel times of one case of tools/cliff_scan.py.  usage: tools/cliff_ktrace.sh N size K dist rlo rhi
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/ck
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/cliff_one.py "$@" > gpurun_out/cliff_one.txt 2>&1
python tools/rocprof_summary.py $OUT gpurun_out/cliff_ktrace.txt > /dev/null
rm -rf $OUT
tail -6 gpurun_out/cliff_one.txt
head -12 gpurun_out/cliff_ktrace.txt | cut -c1-70,90-150
