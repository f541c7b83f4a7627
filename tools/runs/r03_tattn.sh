set -x
timeout 1200 python -m pytest tests/test_gpu_train_hip.py -x -q -s 2>&1 | grep "mfma=2\|passed\|failed\|worst relative\|Error\|error" | tail -30
timeout 1200 python -m pytest tests/test_gpu_module.py -x -q 2>&1 | grep "passed\|failed" | tail -3
timeout 600 python tools/bench_train.py 16 300 6 hip
DIMX_TRAIN_ATTN_VALU=1 timeout 600 python tools/bench_train.py 16 300 3 f32
