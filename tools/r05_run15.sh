#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e15; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or chain" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python tools/r05_gemm_blocks.py $O/plan.json ws72 > $O/run.log 2>&1
python tools/bench_gemm.py --parse $O/trace $O/plan.json > $O/result.txt 2>&1
cat $O/result.txt; rm -rf $O/trace
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export DIMX_NO_WS72=1; else unset DIMX_NO_WS72; fi
  echo "DIMX_NO_WS72=$v" >> $O/bench.txt
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-train-step 2>/dev/null | tail -1 | cut -c1-140 >> $O/bench.txt
done
cat $O/bench.txt
