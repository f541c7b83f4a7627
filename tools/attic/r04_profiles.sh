#!/bin/bash
# Everything under profiles/r04_* comes from this script (one gpurun call on an MI355X):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r04_profiles.sh <commit>'
# then copy gpurun_out/r04/* into profiles/ (tools/pmc_record.py writes profiles/pmc_decode_attn_<hash>.json on the GPU box: see the cp).
set -u
COMMIT=${1:-unknown}
O=gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. headline line (value, parity_mode, roofline, cross_attn_mfma, cross_attn_bundle, train_step, cpu_baseline)
python bench.py --steps 10 --warmup 2 > $O/r04_bench_line.json 2> $O/r04_bench_line.err
# 2. kernel traces: the headline workload, the f32 parity mode
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step > $O/r04_bench_line_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/r04_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32 -- python bench.py --mode f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/r04_parity_line_under_rocprof.json 2>/dev/null
cp $(ls $O/kt32/*/*kernel_stats.csv | head -1) $O/r04_parity_kernel_stats.csv
DIMX_F32_NO_SPLIT=1 python bench.py --mode f32 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r04_parity_line_no_split.json 2>/dev/null
python tools/bench_f32_decode_gemm.py 2>&1 | grep -v amdgpu > $O/r04_f32_decode_gemm.txt
# 3. HBM traffic of the dominant kernel: separate PMC passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python tools/roofline_only.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python tools/roofline_only.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch | grep -A2 "decode_attn\|gemm" > $O/r04_pmc_FETCH_SIZE_roofline_kernels.txt
python tools/pmc_summary.py $O/pmc_write | grep -A2 "decode_attn\|gemm" > $O/r04_pmc_WRITE_SIZE_roofline_kernels.txt
python tools/pmc_record.py $O/pmc_fetch $O/pmc_write $COMMIT > $O/r04_pmc_record.txt 2>&1
cp profiles/pmc_decode_attn_*.json $O/ 2>/dev/null
# 4. prefill attention: timings, ablations, counters (attention_tr.hip)
python tools/bench_attn.py 2>&1 | grep -v amdgpu > $O/r04_attn_bench.txt
DIMX_ATTN_OLD=1 python tools/bench_attn.py 2>&1 | grep -v amdgpu | sed 's/^/round-3 kernel (DIMX_ATTN_OLD=1): /' >> $O/r04_attn_bench.txt
for d in 1 8; do echo "== DIMX_ATTN_DBG=$d (1: no tile DMA after the first, 8: no compute)"; DIMX_ATTN_DBG=$d python tools/bench_attn.py 2>&1 | grep -v amdgpu | grep -v long; done > $O/r04_attn_ablation.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pa1 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pa1 | grep -A9 "attn_tr_kernel" > $O/r04_attn_pmc.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/pa2 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pa2 | grep -A5 "attn_tr_kernel" >> $O/r04_attn_pmc.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pa3 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/pa3 | grep -A5 "attn_tr_kernel" >> $O/r04_attn_pmc.txt
tools/ubench/tr16_probe > $O/r04_tr16_probe.txt 2>&1
# 5. variants of the headline line
DIMX_NO_FUSE_LN0=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/r04_bench_line_no_fuse_ln0.json 2>/dev/null
DIMX_ATTN_OLD=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/r04_bench_line_attn_old.json 2>/dev/null
python bench.py --steps 2 --warmup 1 --samples 10 --no-cpu-baseline --no-parity-mode --no-roofline --no-train-step > $O/r04_bench_samples10.json 2>/dev/null
python bench.py --steps 2 --warmup 1 --batch 64 --frames 1500 --no-cpu-baseline --no-parity-mode --no-roofline --no-train-step > $O/r04_bench_c5_shard.json 2>/dev/null
python bench.py --gpus 2 --steps 1 --warmup 0 > $O/r04_bench_gpus2_on_one_gpu.txt 2>&1; echo "exit code $?" >> $O/r04_bench_gpus2_on_one_gpu.txt
# the 2-rank path with the real model on this 1-GPU box: both ranks on cuda:0, collectives on gloo (a rehearsal, not a measurement)
python bench.py --gpus 2 --rehearse-shared-gpu --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/r04_bench_rehearsal_2ranks_one_gpu.json 2> $O/r04_bench_rehearsal_2ranks_one_gpu.err; echo "exit code $?" >> $O/r04_bench_rehearsal_2ranks_one_gpu.err
# 6. training step
python tools/bench_train.py 16 300 5 all 2>&1 | grep -v amdgpu > $O/r04_train_step.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python tools/bench_train.py 16 300 3 bf16 > /dev/null 2>&1
cp $(ls $O/kt_train/*/*kernel_stats.csv | head -1) $O/r04_train_step_kernel_stats.csv
bash tools/scale_check.sh 1 > $O/r04_scale_check_n1.txt 2>&1
# 7. the other models' training steps (SLM pre-training, legacy generator) next to their PyTorch-autograd restatements
python tools/bench_train_slm.py 16 300 5 all 2>&1 | grep -v amdgpu > $O/r04_train_step_other_models.txt
python tools/bench_train_slm.py 4 300 5 all 0 2>&1 | grep -v amdgpu >> $O/r04_train_step_other_models.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_slm -- python tools/bench_train_slm.py 16 300 3 slm 0 > /dev/null 2>&1
cp $(ls $O/kt_slm/*/*kernel_stats.csv | head -1) $O/r04_train_slm_kernel_stats.csv
python tools/bench_dw_split.py 2>&1 | grep -v amdgpu > $O/r04_dw_split.txt
# 8. the fused feed-forward sublayer: kernel time per ablation (DIMX_MLP_ABL bits: 1 no DMA in the loop, 2 no GELU, 4 no chunk loop,
#    8 no fragment reads, 16 no MFMAs), the headline without it, everything outside the decode loop in situ
for abl in 0 1 2 3 4 11 19; do
  DIMX_MLP_ABL=$abl rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_mlp$abl -- python tools/bench_mlp_fused.py > /dev/null 2>&1
  echo "== DIMX_MLP_ABL=$abl"; python tools/kstat.py $O/kt_mlp$abl mlp_fused; rm -rf $O/kt_mlp$abl
done > $O/r04_mlp_fused.txt
DIMX_NO_FUSED_MLP=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/r04_bench_line_no_fused_mlp.json 2>/dev/null
bash tools/prefill_in_situ.sh > /dev/null 2>&1
cp gpurun_out/insitu/gemm_in_situ.txt $O/r04_prefill_gemm_in_situ.txt; cp gpurun_out/insitu/other_kernels.txt $O/r04_forward_kernels_in_situ.txt
rm -rf $O/kt $O/kt32 $O/kt_train $O/kt_slm $O/pmc_fetch $O/pmc_write $O/pa1 $O/pa2 $O/pa3
ls -la $O
tail -c 400 $O/r04_bench_line.json
