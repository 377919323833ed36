#!/bin/bash
# On the GPU box: SQ counters of every voge kernel of the bench frame, four counters per pass
# (separate rocprofv3 --pmc runs, kernel-trace domain only) -> gpurun_out/refresh/<tag>_pmc_sq_counters.txt
TAG=${1:-r1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/refresh
mkdir -p $OUT
B="python3 bench.py --no-graph --no-cpu-baseline --steps 6 --warmup 2"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/sq_$i -- $B > /dev/null 2>&1
done
python tools/pmc_summary.py $OUT/sq_1 $OUT/sq_2 $OUT/sq_3 $OUT/sq_4 > $OUT/${TAG}_pmc_sq_counters.txt
rm -rf $OUT/sq_1 $OUT/sq_2 $OUT/sq_3 $OUT/sq_4
grep -A17 "trace_fwd_kernel" $OUT/${TAG}_pmc_sq_counters.txt | head -20
