#!/bin/bash
# the prefill MFMA-busy record alone (the full script's pass ran with the best-of-10 leg inside and produced no counter file)
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/pmc_pre
DIMX_PREFILL_GROUPS=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_pre -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline --no-best-of-n > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_pre | grep -A8 "mlp_fused_kernel\|attn_tr_kernel" > $O/r06_pmc_mfma_prefill_kernels.txt
python tools/pmc_prefill_record.py $O/r06_pmc_mfma_prefill_kernels.txt be9cf8f
cp profiles/pmc_prefill_mfma_*.json $O/
rm -rf $O/pmc_pre
python bench.py --steps 10 --warmup 2 > $O/r06_bench_line.json 2> $O/r06_bench_line.err
tail -c 300 $O/r06_bench_line.json
