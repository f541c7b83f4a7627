#!/bin/bash
# profiles/r06_bench_*, r06_*_kernel_stats.csv, r06_pmc_* and the pmc_*.json records of the round's final kernel sources come from this script,
# one gpurun call (the round's experiment files r06_xcdmask_probe / r06_halfchip_engines / r06_x3_gemm / r06_layer_weights_abl have their own scripts
# under tools/runs/):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r06_profiles.sh <commit>'
# then copy gpurun_out/r06/* into profiles/.  The script FAILS (exit 3) when a kernel source has no PMC record afterwards.
set -u
COMMIT=${1:-unknown}
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. HBM traffic of the dominant kernel (xcd_layer_kernel) and of the standalone cross-attention kernel: separate PMC passes
#    (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), folded into profiles/pmc_*_<source hash>.json BEFORE the headline line reads them
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python tools/roofline_only.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python tools/roofline_only.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch | grep -A2 "decode_attn\|xcd_layer\|gemm" > $O/r06_pmc_FETCH_SIZE_roofline_kernels.txt
python tools/pmc_summary.py $O/pmc_write | grep -A2 "decode_attn\|xcd_layer\|gemm" > $O/r06_pmc_WRITE_SIZE_roofline_kernels.txt
python tools/pmc_record.py $O/pmc_fetch $O/pmc_write $COMMIT > $O/r06_pmc_record.txt 2>&1
# 2. MFMA counters of the cross-attention K/V projection (gemm256) -> profiles/pmc_gemm256_<hash>.json
cat > /tmp/xkv.py <<PY
import sys, torch
sys.path.insert(0, ".")
import dimx
from dimx import roofline
print(roofline.cross_kv_gemm(256, 300, "bf16", torch.device("cuda:0"), iters=6))
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python /tmp/xkv.py > $O/r06_cross_kv_line_under_pmc.txt 2>/dev/null
python tools/pmc_summary.py $O/pmc_mfma | grep -A9 "gemm256p2" > $O/r06_pmc_mfma_cross_kv.txt
python tools/pmc_gemm256_record.py $O/r06_pmc_mfma_cross_kv.txt $O/r06_cross_kv_line_under_pmc.txt $COMMIT >> $O/r06_pmc_record.txt 2>&1
# MFMA busy of the two other MFMA kernels of the prefill (fused feed-forward sublayer, prefill attention), same counters
# (one batch on one stream, DIMX_PREFILL_GROUPS=1: with the clip groups of the default path several of these kernels run at the same time
#  and a kernel's counters are diluted by its neighbours -- 19 % instead of 27 % MFMA busy for the fused feed-forward kernel)
DIMX_PREFILL_GROUPS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_pre -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline --no-best-of-n > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_pre | grep -A8 "mlp_fused_kernel\|attn_tr_kernel" > $O/r06_pmc_mfma_prefill_kernels.txt
python tools/pmc_prefill_record.py $O/r06_pmc_mfma_prefill_kernels.txt $COMMIT >> $O/r06_pmc_record.txt 2>&1
cp profiles/pmc_*.json $O/ 2>/dev/null
MISSING=0
python - <<PY || MISSING=1
import os, sys
sys.path.insert(0, ".")
import dimx
from dimx import roofline as R
need = ["pmc_layer_chain_%s.json" % R.kernel_source_hash(R.LAYER_CHAIN_SOURCES), "pmc_decode_attn_%s.json" % R.kernel_source_hash(),
        "pmc_gemm256_%s.json" % R.kernel_source_hash("gemm256.hip"), "pmc_prefill_mfma_%s.json" % R.kernel_source_hash(R.PREFILL_MFMA_SOURCES)]
miss = [n for n in need if not os.path.exists(os.path.join(R.PROFILES, n))]
print("PMC records:", "all present" if not miss else "MISSING " + ", ".join(miss))
sys.exit(1 if miss else 0)
PY
# 3. headline line (value, parity_mode, bf16_vs_f32, roofline incl. phases, cross_attn_mfma, cross_attn_bundle, train_step, cpu_baseline)
python bench.py --steps 10 --warmup 2 > $O/r06_bench_line.json 2> $O/r06_bench_line.err
# 4. kernel traces: the headline workload, the f32 parity mode, the C5 shard
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step --no-best-of-n > $O/r06_bench_line_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/r06_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32 -- python bench.py --mode f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/r06_parity_line_under_rocprof.json 2>/dev/null
cp $(ls $O/kt32/*/*kernel_stats.csv | head -1) $O/r06_parity_kernel_stats.csv
python bench.py --steps 2 --warmup 1 --batch 64 --frames 1500 --no-cpu-baseline --no-parity-mode --no-train-step > $O/r06_bench_c5_shard.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktc5 -- python bench.py --steps 2 --warmup 1 --batch 64 --frames 1500 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline --no-best-of-n > /dev/null 2>&1
cp $(ls $O/ktc5/*/*kernel_stats.csv | head -1) $O/r06_c5_kernel_stats.csv
# 5. variants (same box): the parity mode with the exact-f32 MFMA decode GEMMs, the bf16 mode without the layer kernel
DIMX_NO_X3=1 python bench.py --mode f32 --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-roofline --no-mode-compare > $O/r06_parity_line_no_x3.json 2>/dev/null
python bench.py --mode f32 --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-mode-compare > $O/r06_parity_line.json 2>/dev/null
DIMX_NO_LAYER_CHAIN=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/r06_bench_line_no_layer_kernel.json 2>/dev/null
DIMX_LAYER_PROF=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity-mode --no-train-step 2>&1 | grep -A13 "layer-kernel stamps" > $O/r06_layer_kernel_stamps_last_step.txt
python bench.py --steps 2 --warmup 1 --samples 10 --no-cpu-baseline --no-parity-mode --no-roofline --no-train-step > $O/r06_bench_samples10.json 2>/dev/null
# 7. training step (unchanged kernels; the line's train_step object) and the N = 1 leg of the scale check
python tools/bench_train.py 16 300 5 all 2>&1 | grep -v amdgpu > $O/r06_train_step.txt
bash tools/scale_check.sh 1 > $O/r06_scale_check_n1.txt 2>&1
rm -rf $O/kt $O/kt32 $O/ktc5 $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_pre
ls -la $O
tail -c 600 $O/r06_bench_line.json
[ $MISSING = 0 ] || { echo "FAILED: a kernel source has no PMC record (see above)"; exit 3; }
