set -x
timeout 900 python -m pytest tests/test_gpu_train_hip.py -q -x -s 2>&1 | grep -v "^$" | tail -60
