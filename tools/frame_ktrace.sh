#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_frame -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-graph --no-variants --only-stage frame > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_frame | cut -c1-70,75-150 | head -24
rm -rf gpurun_out/prof_frame
