#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e10; mkdir -p $O; rm -f $O/result.txt
for kt in 1 2 4; do
  echo "== DIMX_DEC_KT=$kt" >> $O/result.txt
  DIMX_DEC_KT=$kt timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "fragment_packed" 2>&1 | tail -2 >> $O/result.txt
  for abl in 0 4 8; do
    echo "-- DIMX_DEC_ABL=$abl" >> $O/result.txt
    DIMX_DEC_KT=$kt DIMX_DEC_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python tools/r05_gemm_blocks.py $O/plan.json frag > $O/run.log 2>&1
    python tools/bench_gemm.py --parse $O/trace $O/plan.json >> $O/result.txt 2>&1
    rm -rf $O/trace
  done
done
cat $O/result.txt
