#!/bin/bash
# On the GPU box: the VOGE_XCD_CHAIN experiment (trace_bin.h: a region's binA slices and its binB quads on one XCD, so that binB
# could find binA's segments in that XCD's L2).  Frame rate + entry time interleaved (tools/ab_bench.sh), the per-kernel
# durations (kernel trace), and the FETCH_SIZE / WRITE_SIZE of binA / binB / the sweep for both builds.
# usage: tools/xcd_chain_ab.sh      (needs build/variants/xcdchain.so: tools/tune_variants.sh xcdchain:"-DVOGE_XCD_CHAIN=1")
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/xcd; rm -rf $OUT; mkdir -p $OUT
V=$ROOT/build/variants/xcdchain.so
for CFG in cfg3_50k_512 cfg4_200k_1024; do
  echo "==== $CFG"
  for r in 1 2 3; do
    for v in "" "$V"; do
      VOGE_HIP_LIB=$v python bench.py --no-cpu-baseline --no-variants --config $CFG --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('lib=%-10s frame %.1f fps  entry %.1f us  lean %.1f us' % (sys.argv[1] and 'xcdchain' or 'in-tree', d['value'], d['roofline']['avg_launch_ms']*1e3, s['trace_lean_fwd']['ms']*1e3))" "$v"
    done
  done
  B="python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 30 --warmup 5 --config $CFG --only-stage trace_fwd"
  for v in "" "$V"; do
    export VOGE_HIP_LIB=$v
    echo "-- ${v:+xcdchain}${v:-in-tree}: kernel durations, then FETCH_SIZE / WRITE_SIZE"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > /dev/null 2>&1
    python tools/rocprof_summary.py $OUT/kt $OUT/kt.txt > /dev/null; grep -E "binA|binB|sweep|trace_fwd" $OUT/kt.txt | cut -c1-40,72-140; rm -rf $OUT/kt
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -- $B > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw -- $B > /dev/null 2>&1
    python tools/pmc_summary.py $OUT/pf $OUT/pw 2>/dev/null | grep -E -A2 "binA|binB|sweep_iso" | cut -c1-110; rm -rf $OUT/pf $OUT/pw
  done
  unset VOGE_HIP_LIB
done
