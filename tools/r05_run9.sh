#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e9; mkdir -p $O
for abl in 0 32 8 24 7 4 3; do
  echo "== DIMX_DEC_ABL=$abl (1 no W refills, 2 no A refills, 4 no LDS reads / MFMAs, 8 one k-tile, 16 no stores, 32 empty kernel)" >> $O/result.txt
  DIMX_DEC_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d $O/trace$abl -o g -- python tools/r05_gemm_blocks.py $O/plan.json frag > $O/run.log 2>&1
  python tools/bench_gemm.py --parse $O/trace$abl $O/plan.json >> $O/result.txt 2>&1
  rm -rf $O/trace$abl
done
cat $O/result.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tracew -o g -- python tools/r05_gemm_blocks.py $O/plan.json ws72 > $O/run.log 2>&1
python tools/bench_gemm.py --parse $O/tracew $O/plan.json > $O/result_ws72.txt 2>&1
cat $O/result_ws72.txt
rm -rf $O/tracew
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "fragment_packed" 2>&1 | tail -3
