#!/bin/bash
# On the GPU box: bench.py's frame rate and trace-entry time for several builds of the library, interleaved.
# usage: tools/ab_bench.sh rounds lib [lib ...]      ("" = the in-tree build)
R=$1; shift
for r in $(seq $R); do
  for v in "$@"; do
    VOGE_HIP_LIB=$v python bench.py --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('lib=%-32s frame %.1f fps  entry %.1f us (frac %.4f)  lean %.1f us  fragment_bwd %.1f us  fragments_fwd %.1f us' % (sys.argv[1] or 'in-tree', d['value'], d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac'], s['trace_lean_fwd']['ms']*1e3, s['fragment_bwd']['ms']*1e3, s['fragments_fwd']['ms']*1e3))" "$v"
  done
done
