set -x
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6
timeout 600 python -m pytest tests/test_gpu_train_hip.py -q -s -k "gradients or backward or matches_autograd" 2>&1 | grep "worst relative\|passed\|failed" | head -5
timeout 600 python __graft_entry__.py smoke | tail -3
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-mode | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d['cross_attn_mfma']['util_pct'], d['cross_attn_mfma']['pmc_mfma_busy_pct'], d['train_step']['ms_per_step'])"
