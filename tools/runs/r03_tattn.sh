set -x
timeout 1200 python -m pytest tests/test_gpu_train_hip.py tests/test_gpu_module.py tests/test_gpu_slm.py tests/test_gpu_legacy.py -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8
