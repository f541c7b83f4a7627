#!/bin/bash
# full -m gpu suite + a short bench line (round 4 checkpoints)
set -u
O=gpurun_out/r04c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > $O/bench_line.json 2> $O/bench_line.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench_line.json"))
print({k: d[k] for k in ("value","ms_per_step","chain_faults")}, d.get("parity_mode",{}).get("value"), d.get("train_step",{}).get("ms_per_step"))
print(d["roofline"]["frac"], d["cross_attn_mfma"]["util_pct"], d.get("cross_attn_bundle",{}).get("util_pct"), d.get("cross_attn_bundle",{}).get("teacher_forced_cross_attention"))
PY
tail -3 $O/bench_line.err
