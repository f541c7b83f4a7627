set -x
DIMX_G256_VAR=5 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm256" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_s2s.py -q -x -s -k "chain_fault or decode_tf" 2>&1 | grep -v "^$" | tail -12
G256_PROF_TOO=0 timeout 1200 python tools/g256_var.py 3 2>&1 | grep -v amdgpu
