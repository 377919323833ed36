#!/bin/bash
# On the GPU box: every kernel the GPU test-suite launches, by average duration -- small inputs, so anything long is a path nobody tuned.
# usage: tools/tests_ktrace.sh [pytest args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/tktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 -m pytest "${@:-tests}" -x -q -m gpu -k "not bench and not demo and not stress" > $OUT/out.txt 2>&1
tail -2 $OUT/out.txt
python tools/rocprof_summary.py $OUT gpurun_out/tests_ktrace.txt > /dev/null
rm -rf $OUT
grep voge gpurun_out/tests_ktrace.txt | sort -k5 -n -r -t$'\t' | awk '{print}' | cut -c1-90,100-160 | sort -k4 -n -r | head -40
