set -x
timeout 900 python -m pytest tests/test_gpu_train_hip.py tests/test_gpu_module.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_legacy.py -x -q -s -k "train_epoch" 2>&1 | tail -8
timeout 600 python tools/bench_train.py 16 300 5 all
