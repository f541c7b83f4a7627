set -x
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" 2>&1 | tail -3
G256_PROF_TOO=0 timeout 900 python tools/g256_var.py 3 2>&1 | grep -v amdgpu | grep -v "^round"
echo "== prefill shapes, default eligibility"; timeout 300 python tools/bench_prefill.py 0 2>&1 | grep -v amdgpu
echo "== prefill shapes, DIMX_G256_ALL=1"; DIMX_G256_ALL=1 timeout 300 python tools/bench_prefill.py 0 2>&1 | grep -v amdgpu
