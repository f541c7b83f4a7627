set -x
timeout 900 python -m pytest tests/test_gpu_train_hip.py -q -x -s 2>&1 | grep -v "^$" | tail -30
echo "== prefill after eligibility change"; timeout 300 python tools/bench_prefill.py 0 2>&1 | grep -v amdgpu
