#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/pmcx
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcx -- python tools/r06_x3_gemm_bench.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pmcx | grep -A9 "gemm_x3_kernel\|gemm_ws_kernel<float" > $O/x3_pmc.txt
rm -rf $O/pmcx
cat $O/x3_pmc.txt | head -80
