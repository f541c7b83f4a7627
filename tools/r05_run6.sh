#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/e6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "fragment_packed or one_block_per_cu or chain_deferred" 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
{
for args in "73 1152 4608 4 0" "73 4608 1152 0 3" "73 2304 1152 2 0"; do
  python tools/r05_gemm_stamps.py $args 2>&1 | grep -v amdgpu.ids
done
} > $O/stamps.txt 2>&1
cat $O/stamps.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o g -- python tools/r05_gemm_blocks.py $O/plan.json ws72 > $O/run.log 2>&1
python tools/bench_gemm.py --parse $O/trace $O/plan.json > $O/result.txt 2>&1
cat $O/result.txt
rm -rf $O/trace
