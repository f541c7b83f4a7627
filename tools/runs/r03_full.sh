set -x
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d['cross_attn_mfma']['util_pct'], d['cross_attn_mfma']['pmc_mfma_busy_pct'])"
