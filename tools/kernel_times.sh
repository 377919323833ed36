#!/bin/bash
# usage: kt.sh variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VOGE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/$1.so
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > /dev/null 2>&1
echo "== $1"; python tools/rocprof_summary.py gpurun_out/prof_$1 | cut -c1-150 | grep -E "bin|trace_fwd|prep|tile_order"
