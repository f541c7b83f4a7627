cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p1 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/p1 | grep -A9 "attn_tr_kernel<64"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/p2 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/p2 | grep -A9 "attn_tr_kernel<64"
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/p3 -- python tools/bench_attn.py > /dev/null 2>&1
python tools/pmc_summary.py $O/p3 | grep -A9 "attn_tr_kernel<64"
rm -rf $O/p1 $O/p2 $O/p3
