cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in 0 14; do
 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_g${cfg}c -- python tools/gemm_pmc.py 76800 1536 1152 $cfg > /dev/null 2>&1
 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_g${cfg}d -- python tools/gemm_pmc.py 76800 1536 1152 $cfg > /dev/null 2>&1
 echo "== cfg $cfg"; python tools/pmc_summary.py gpurun_out/pmc_g${cfg}c | grep -A5 "gemm"; python tools/pmc_summary.py gpurun_out/pmc_g${cfg}d | grep -A2 "gemm"
done
for ng in 1 2 3; do echo "== n_group $ng"; DIMX_G256_NGROUP=$ng python tools/bench_prefill.py 0 2>&1 | grep -v amdgpu | head -1; done
