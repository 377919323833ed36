#!/bin/bash
# On the GPU box: per-kernel times of the bench frame with [N,3,3] L L^T sigmas (eager launches).  usage: tools/aniso_ktrace.sh [lib]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VOGE_HIP_LIB=$1
rm -rf gpurun_out/prof_aniso
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_aniso -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-variants --only-stage frame --anisotropic > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_aniso 2>/dev/null | cut -c1-70,100-150 | sed -n 1,12p
rm -rf gpurun_out/prof_aniso
