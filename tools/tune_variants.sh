#!/bin/bash
# Build tuning variants of libvoge_hip.so (compile-time constants via -D) into build/variants/.
# usage: tools/tune_variants.sh name1:"-DVOGE_BWD_TH=2 -DVOGE_BWD_NE=64" name2:"..." ...
# Run them on the GPU box with tools/tune_run.sh (VOGE_HIP_LIB selects the build).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants
mkdir -p "$OUT"
cd "$ROOT/voge_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -I$ROOT/include -I."
pids=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -shared -o "$OUT/$name.so" trace_fwd.hip trace_bwd.hip composite.hip merge_blend.hip rays.hip extras.hip fragment_bwd.hip &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
ls -la "$OUT"
