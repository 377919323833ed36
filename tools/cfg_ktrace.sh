#!/bin/bash
# On the GPU box: per-kernel durations of the bench frame of a config (eager launches).  usage: tools/cfg_ktrace.sh [config] [extra bench flags]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${1:-cfg4_200k_1024}; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-variants --config $CFG "$@" > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_cfg | cut -c1-70,100-150 | head -${LINES_OUT:-22}
rm -rf gpurun_out/prof_cfg
