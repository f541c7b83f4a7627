#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e11; mkdir -p $O; rm -f $O/result.txt
for abl in 0 64 4; do
    echo "-- DIMX_DEC_ABL=$abl (64: consumer waves 4, 5 skip their reads and MFMAs: one computing wave per SIMD)" >> $O/result.txt
    DIMX_DEC_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python tools/r05_gemm_blocks.py $O/plan.json frag > $O/run.log 2>&1
    python tools/bench_gemm.py --parse $O/trace $O/plan.json >> $O/result.txt 2>&1
    rm -rf $O/trace
done
cat $O/result.txt
