set -x
timeout 900 python -m pytest tests/test_gpu_train_hip.py -x -q -s 2>&1 | grep -v "^$" | tail -60
timeout 600 python tools/bench_train.py 16 300 5 hip
DIMX_TRAIN_ATTN_VALU=1 timeout 600 python tools/bench_train.py 16 300 3 bf16
