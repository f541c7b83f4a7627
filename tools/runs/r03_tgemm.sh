timeout 900 python tools/bench_train_gemm.py 0 14 34 4 2>&1 | grep -v amdgpu
