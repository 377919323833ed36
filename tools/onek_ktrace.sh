#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in "$@"; do
  OUT=gpurun_out/tk1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$k" > $OUT/out.txt 2>&1
  python tools/rocprof_summary.py $OUT gpurun_out/tk1.txt > /dev/null
  echo "== $k: $(tail -1 $OUT/out.txt)"; grep "sweep_iso_kernel<true>\|binB_kernel<true>" gpurun_out/tk1.txt | cut -c1-60,100-160
  rm -rf $OUT
done
