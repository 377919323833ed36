#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_aniso -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-variants --anisotropic > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_aniso | cut -c1-70,100-150 | head -24
rm -rf gpurun_out/prof_aniso
