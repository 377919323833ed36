#!/bin/bash
# On the GPU box: dynamic instruction counts of fragment_bwd_kernel for every build/variants/*.so (ablation builds:
# tools/tune_variants.sh abl1:"-DVOGE_FB_ABL=1" ...; bit 0 no table, 1 no composite, 2 no colour gathers, 3 no record gathers)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/fbvalu
rm -rf $OUT; mkdir -p $OUT
for lib in build/variants/*.so; do
  name=$(basename $lib .so)
  export VOGE_HIP_LIB=$ROOT/$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/$name -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 6 --warmup 2 --only-stage frame > /dev/null 2>&1
  echo "== $name"
  python tools/pmc_summary.py $OUT/$name | grep -A5 "fragment_bwd_kernel<0, 3, 2, unsigned int, true, true>"
  rm -rf $OUT/$name
done
