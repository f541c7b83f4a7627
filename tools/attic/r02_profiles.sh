#!/bin/bash
# Everything under profiles/r02_* comes from this script (one gpurun call on an MI355X):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r02_profiles.sh <commit>'
# then copy gpurun_out/r02/* into profiles/ (tools/pmc_record.py writes profiles/pmc_decode_attn_<hash>.json itself,
# into gpurun_out/r02 on the GPU box: see the cp below).
set -u
COMMIT=${1:-unknown}
O=gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. headline line (value, parity_mode, roofline, cross_attn_mfma, cpu_baseline)
python bench.py > $O/r02_bench_line.json 2> $O/r02_bench_line.err
# 2. kernel trace of the same workload
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode > $O/r02_bench_line_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/r02_bench_kernel_stats.csv
# 3. HBM traffic of the dominant kernel: separate PMC passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python tools/roofline_only.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python tools/roofline_only.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch | grep -A2 "decode_attn\|gemm" > $O/r02_pmc_FETCH_SIZE_roofline_kernels.txt
python tools/pmc_summary.py $O/pmc_write | grep -A2 "decode_attn\|gemm" > $O/r02_pmc_WRITE_SIZE_roofline_kernels.txt
python tools/pmc_record.py $O/pmc_fetch $O/pmc_write $COMMIT > $O/r02_pmc_record.txt 2>&1
cp profiles/pmc_decode_attn_*.json $O/ 2>/dev/null
# 4. MFMA counters of the cross-attention K/V projection (the fused 4-layer launch of gemm256)
cat > /tmp/xkv.py <<PY
import sys, torch
sys.path.insert(0, ".")
import dimx
from dimx import roofline
print(roofline.cross_kv_gemm(256, 300, "bf16", torch.device("cuda:0"), iters=6))
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python /tmp/xkv.py > $O/r02_cross_kv_line_under_pmc.txt 2>/dev/null
python tools/pmc_summary.py $O/pmc_mfma | grep -A9 "gemm256" > $O/r02_pmc_mfma_cross_kv.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- python /tmp/xkv.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_l2 | grep -A5 "gemm256" >> $O/r02_pmc_mfma_cross_kv.txt
python tools/pmc_gemm256_record.py $O/r02_pmc_mfma_cross_kv.txt $O/r02_cross_kv_line_under_pmc.txt $COMMIT >> $O/r02_pmc_record.txt 2>&1
cp profiles/pmc_gemm256_*.json $O/ 2>/dev/null
# 5. phase breakdown of the chain kernels, prefill GEMM table, f32 / other variants
python tools/chain_phases.py 2>&1 | grep -v amdgpu > $O/r02_chain_phases.txt
DIMX_GEMM_PROF=1 python tools/gemm_phases.py 34 2>&1 | grep -v amdgpu > $O/r02_gemm_phases.txt
DIMX_GEMM_PROF=1 python tools/gemm_phases.py 34 ln 2>&1 | grep -v amdgpu >> $O/r02_gemm_phases.txt
for c in 3 34; do DIMX_GEMM_CFG_SMALL=$c python tools/gemm_ab.py 2>&1 | tail -1; done > $O/r02_gemm_ab.txt
tools/ubench/cu_load_rate > $O/r02_cu_load_rate.txt 2>&1
DIMX_NO_DEFER_LN=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline > $O/r02_bench_line_no_defer_ln.json 2>/dev/null
python tools/bench_prefill.py 0 14 2>&1 | grep -v amdgpu > $O/r02_prefill_gemm.txt
DIMX_NO_CHAIN=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-roofline > $O/r02_bench_line_no_chain.json 2>/dev/null
python bench.py --steps 2 --warmup 1 --samples 10 --no-cpu-baseline --no-parity-mode --no-roofline > $O/r02_bench_samples10.json 2>/dev/null
python bench.py --steps 2 --warmup 1 --batch 64 --frames 1500 --no-cpu-baseline --no-parity-mode --no-roofline > $O/r02_bench_c5_shard.json 2>/dev/null
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_l2
ls -la $O
tail -c 600 $O/r02_bench_line.json
