#!/bin/bash
# ablations + counters of attention_tr.hip (tuning; DIMX_ATTN_DBG bits: 1 no in-loop tile DMA, 2 no exp, 4 no P.V, 8 no compute,
# 16 no barrier, 32 no stores; DIMX_ATTN_BPC: persistent blocks per CU)
set -u
O=gpurun_out/r04b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -k "row_major" 2>&1 | grep -E "passed|failed|FAILED" | tee $O/pytest.txt
for d in ${ABL:-0 1 2 4 8 16 32 9 63}; do echo "== DBG $d"; DIMX_ATTN_DBG=$d python tools/bench_attn.py 2>&1 | grep -v amdgpu | grep -v long; done | tee $O/abl.txt
for n in ${BPC:-1 2 4}; do echo "== BPC $n"; DIMX_ATTN_BPC=$n python tools/bench_attn.py 2>&1 | grep -v amdgpu | grep -v long; done | tee $O/bpc.txt
if [ -n "${PMC:-1}" ]; then
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc1 | grep -A9 attn_tr | tee $O/pmc1.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc2 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc2 | grep -A9 attn_tr | tee $O/pmc2.txt
rm -rf $O/pmc1 $O/pmc2
fi
