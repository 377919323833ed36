#!/bin/bash
# On the GPU box: A/B of build variants on the bench frame's binning + sweep kernels.  usage: tools/ab_ktrace.sh variant...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  if [ "$v" = "head" ]; then unset VOGE_HIP_LIB; else export VOGE_HIP_LIB=$ROOT/build/variants/$v.so; fi
  echo "== $v"
  bash $ROOT/tools/frame_ktrace.sh 2>&1 | grep -E "trace_fwd|binB|binA" | cut -c1-40,72-130
done
