#!/bin/bash
# Device assembly of one csrc unit (gfx950): tools/asm.sh trace_fwd [-DFLAG ...] -> /tmp/<unit>.s ; resource lines printed
U=$1; shift
cd "$(dirname "$0")/../voge_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I../../include -I. \
  -S --offload-device-only "$@" -o /tmp/$U.s $U.hip 2>&1 | grep -v "hip-link\|^$"
grep -E "^\s+\.(name|vgpr_count|sgpr_count|group_segment_fixed_size):" /tmp/$U.s | paste - - - - | sed 's/  */ /g' | cut -c1-200
