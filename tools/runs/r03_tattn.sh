set -x
timeout 1200 python -m pytest tests/test_gpu_train_hip.py -x -q -s 2>&1 | grep "train attention.*mfma=1\|passed\|failed" | tail -24
timeout 600 python tools/bench_train.py 16 300 8 bf16
