#!/bin/bash
# On the GPU box: the frame's kernel timeline (durations and the gaps between launches), eager and under HIP graph replay.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/fg; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/e -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 200 --warmup 5 --only-stage frame > /dev/null 2>&1
python tools/frame_gaps.py $OUT/e "eager"
rocprofv3 --kernel-trace --output-format csv -d $OUT/g -- python3 bench.py --no-cpu-baseline --no-variants --steps 200 --warmup 5 --only-stage frame > /dev/null 2>&1
python tools/frame_gaps.py $OUT/g "hip graph replay"
rm -rf $OUT
