#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/e4/trace -o g -- python tools/r05_gemm_blocks.py gpurun_out/e4/plan.json ws72 > gpurun_out/e4/run.log 2>&1
python tools/bench_gemm.py --parse gpurun_out/e4/trace gpurun_out/e4/plan.json > gpurun_out/e4/result.txt 2>&1
cat gpurun_out/e4/result.txt
rm -rf gpurun_out/e4/trace
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export DIMX_NO_WS72=1; else unset DIMX_NO_WS72; fi
  echo "DIMX_NO_WS72=$v" >> gpurun_out/e4/bench.txt
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/e4/bench.txt
done
cat gpurun_out/e4/bench.txt
