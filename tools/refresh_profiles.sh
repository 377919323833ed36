#!/bin/bash
# On the GPU box: regenerate everything under profiles/ for the current build (outputs land in
# gpurun_out/refresh/, which gpurun merges back; copy them into profiles/ locally afterwards).
# usage: tools/refresh_profiles.sh <round-tag, e.g. r2>
TAG=${1:-r5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
for CFG in cfg3_50k_512 cfg4_200k_1024 cfg5_shapefit_128; do
  # pass 1: nothing but the stand-alone trace entry point (the roofline's kernel chain, act / dsd included);
  # pass 2: the frame as bench.py launches it (scalar sigmas: fragments without act / dsd, fused backward)
  B="python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 10 --warmup 3 --config $CFG"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --only-stage trace_fwd > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --only-stage trace_fwd > /dev/null 2>&1
  python tools/traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_traffic.json $CFG
  rm -rf $OUT/pmc_fetch $OUT/pmc_write
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --only-stage frame > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --only-stage frame > /dev/null 2>&1
  python tools/traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_traffic.json $CFG frame
  if [ $CFG = cfg3_50k_512 ]; then python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/${TAG}_pmc_fetch_write_size.txt 2>/dev/null; fi
  rm -rf $OUT/pmc_fetch $OUT/pmc_write
done
cp $OUT/${TAG}_traffic.json profiles/${TAG}_traffic.json     # bench.py reads it for roofline.traffic
# kernel traces, one per entry point so that every figure of the bench line can be recomputed from profiles/ alone:
#  _trace_entry: nothing but the stand-alone voge_trace_topk_fwd_iso (the roofline's chain: binA + binB + trace_fwd_kernel)
#  _frame:       nothing but the frame's steps (fragments forward, shade forward, fused backward)
#  _summary:     the whole default command (frame + every stand-alone stage timing), as in rounds 1-2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 30 --warmup 5 --only-stage trace_fwd > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/ktrace $OUT/${TAG}_kernel_trace_trace_entry.txt > /dev/null
rm -rf $OUT/ktrace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 100 --warmup 5 --only-stage frame > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/ktrace $OUT/${TAG}_kernel_trace_frame.txt > /dev/null
rm -rf $OUT/ktrace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 bench.py --no-graph --no-cpu-baseline --no-variants --steps 30 --warmup 5 > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/ktrace $OUT/${TAG}_kernel_trace_summary.txt > /dev/null
rm -rf $OUT/ktrace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 bench.py --loop --no-graph --no-cpu-baseline --steps 100 > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/ktrace $OUT/${TAG}_kernel_trace_cfg5_loop.txt > /dev/null
rm -rf $OUT/ktrace
python bench.py | tail -1 > $OUT/${TAG}_bench_line.json
python bench.py --no-graph --no-cpu-baseline --no-variants --steps 300 | tail -1 > $OUT/${TAG}_bench_line_eager.json      # (300 steps: the first frames behind a synchronisation run on a cold host -- 30 eager steps read 0.284 ms where 300 read 0.249)
python bench.py --config cfg4_200k_1024 --no-cpu-baseline --no-variants --steps 20 | tail -1 > $OUT/${TAG}_bench_line_cfg4.json
python bench.py --config cfg5_shapefit_128 --no-variants --steps 50 | tail -1 > $OUT/${TAG}_bench_line_cfg5.json
python bench.py --loop | tail -1 > $OUT/${TAG}_bench_line_cfg5_loop.json
head -c 700 $OUT/${TAG}_bench_line.json; echo; grep -E "voge" $OUT/${TAG}_kernel_trace_summary.txt | cut -c1-60,90-150
