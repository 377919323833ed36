#!/bin/bash
# On the GPU box: per-kernel split of tools/general_path_ab.py's three cases.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for c in iso iso3x3 aniso; do
  OUT=gpurun_out/gab_$c
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/general_path_ab.py $c $1 2>/dev/null | grep "^$c"
  python3 tools/rocprof_summary.py $OUT | grep "voge::" | cut -c1-60,92-150 | head -12
  rm -rf $OUT
done
