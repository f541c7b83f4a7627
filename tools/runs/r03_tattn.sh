set -x
timeout 1200 python -m pytest tests/test_gpu_train_hip.py tests/test_gpu_module.py -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 600 python tools/bench_train.py 16 300 8 bf16
timeout 600 python tools/bench_train.py 16 300 4 f32
