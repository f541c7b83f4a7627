#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e7; mkdir -p $O
{
for abl in 0 1 2 3 4 7; do
  echo "== DIMX_DEC_ABL=$abl"
  DIMX_DEC_ABL=$abl python tools/r05_gemm_stamps.py 73 4608 1152 0 3 2>&1 | grep -v amdgpu.ids
done
} > $O/stamps.txt 2>&1
cat $O/stamps.txt
