#!/bin/bash
# On the GPU box: bench every build/variants/*.so (stage times from bench.py).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for lib in build/variants/*.so; do
  echo "== $lib"
  VOGE_HIP_LIB=$ROOT/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],1), 'fps', ' '.join('%s=%.1f'%(k,v['ms']*1000) for k,v in d['stages'].items()))"
done
