#!/bin/bash
# round 5, GPU call 2: ws72 kernel parity tests, per-shape A/B under a kernel trace, end-to-end A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e2
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or chain" 2>&1 | tail -5 > gpurun_out/e2/tests.txt
cat gpurun_out/e2/tests.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/e2/trace -o g -- python tools/r05_gemm_blocks.py gpurun_out/e2/plan.json ws72 > gpurun_out/e2/run.log 2>&1
python tools/bench_gemm.py --parse gpurun_out/e2/trace gpurun_out/e2/plan.json > gpurun_out/e2/result.txt 2>&1
cat gpurun_out/e2/result.txt
rm -rf gpurun_out/e2/trace
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export DIMX_NO_WS72=1; else unset DIMX_NO_WS72; fi
  echo "DIMX_NO_WS72=$v" >> gpurun_out/e2/bench.txt
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/e2/bench.txt
done
cat gpurun_out/e2/bench.txt
