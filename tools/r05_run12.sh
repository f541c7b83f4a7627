#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e12; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or chain" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
for kt in 1 2; do
  echo "== DIMX_WS72_KT=$kt" >> $O/result.txt
  DIMX_WS72_KT=$kt rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python tools/r05_gemm_blocks.py $O/plan.json ws72 > $O/run.log 2>&1
  python tools/bench_gemm.py --parse $O/trace $O/plan.json >> $O/result.txt 2>&1
  rm -rf $O/trace
done
cat $O/result.txt
for kt in 1 2 1 2; do
  echo "DIMX_WS72_KT=$kt" >> $O/bench.txt
  DIMX_WS72_KT=$kt python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-140 >> $O/bench.txt
done
cat $O/bench.txt
