#!/bin/bash
# On the GPU box: the frame's ms per step for every build/variants/*.so, R interleaved rounds (same box, same minutes).
# usage: tools/tune_ab.sh [rounds] [bench.py arguments ...]
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
R=${1:-2}; shift
for r in $(seq $R); do
  for lib in build/variants/*.so; do
    VOGE_HIP_LIB=$ROOT/$lib timeout 300 python bench.py --no-cpu-baseline --no-variants --no-launch-probe "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('%-14s %.4f ms  (repeats %s)  sweep-entry %.1f lean %.1f shade_fwd %.1f shade_bwd %.1f' % (sys.argv[1].split('/')[-1][:-3], d['ms_per_step'], ' '.join('%.4f' % x for x in d['repeat_ms_per_step']), d['roofline']['avg_launch_ms']*1e3, s.get('trace_lean_fwd',{'ms':0})['ms']*1e3, s.get('frame_shade_fwd',{'ms':0})['ms']*1e3, s.get('frame_shade_bwd',{'ms':0})['ms']*1e3))" $lib
  done
done
