set -x
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_s2s.py tests/test_gpu_vq.py tests/test_gpu_configs.py tests/test_gpu_slm.py tests/test_gpu_legacy.py -x -q 2>&1 | grep "passed\|failed" | tail -3
for e in "DIMX_QKV_VT=1" "X=1" "DIMX_G256_ALL=1" "DIMX_QKV_VT=1" "X=1" "DIMX_G256_ALL=1"; do
echo "== $e"
env $e timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-train-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
