#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e8; mkdir -p $O
for abl in 0 1 2 3 4 5 6 7; do
  echo "== DIMX_DEC_ABL=$abl (1 no W refills, 2 no A refills, 4 no LDS reads / MFMAs)" >> $O/result.txt
  DIMX_DEC_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d $O/trace$abl -o g -- python tools/r05_gemm_blocks.py $O/plan.json frag > $O/run.log 2>&1
  python tools/bench_gemm.py --parse $O/trace$abl $O/plan.json >> $O/result.txt 2>&1
  rm -rf $O/trace$abl
done
cat $O/result.txt
