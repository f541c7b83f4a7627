#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e3
{
for args in "34 1024 4608 4 0" "34 1152 4608 4 0" "72 1152 4608 4 0" "34 4096 1152 0 3" "34 4608 1152 0 3" "72 4608 1152 0 3" "34 2304 1152 2 0" "72 2304 1152 2 0"; do
  python tools/r05_gemm_stamps.py $args
done
} > gpurun_out/e3/stamps.txt 2>&1
cat gpurun_out/e3/stamps.txt
