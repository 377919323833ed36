#!/bin/bash
# On the GPU box: the frame's three entry points (stage timings of bench.py) for every build/variants/*.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for lib in build/variants/*.so; do
  for st in trace_lean_fwd composite_shade_fwd fragment_bwd; do
    VOGE_HIP_LIB=$ROOT/$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --only-stage $st "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', ' '.join('%s=%.1f us (same buffers %.1f)'%(k,v['ms']*1000,v['ms_same_buffers']*1000) for k,v in d['stages'].items()))"
  done
done
