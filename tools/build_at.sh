#!/bin/bash
# Build libvoge_hip.so as it was at a git commit into build/variants/<name>.so (for interleaved A/B runs: tools/ab_bench.sh).
# usage: tools/build_at.sh <commit> <name> [extra -D flags]
set -e
C=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/build/src_$NAME
rm -rf "$SRC"; mkdir -p "$SRC" "$ROOT/build/variants"
git -C "$ROOT" archive "$C" voge_amd/csrc include | tar -x -C "$SRC"
cd "$SRC/voge_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function \
  -I"$SRC/include" -I. "$@" -shared -o "$ROOT/build/variants/$NAME.so" trace_fwd.hip trace_bwd.hip composite.hip merge_blend.hip rays.hip extras.hip fragment_bwd.hip
rm -rf "$SRC"
ls -la "$ROOT/build/variants/$NAME.so"
