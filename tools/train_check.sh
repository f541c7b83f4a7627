#!/bin/bash
# training step: gradient-parity tests, step time, launches per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/train; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_hip.py -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -5
python tools/bench_train.py 16 300 8 bf16 2>&1 | grep -v amdgpu
python tools/bench_train.py 4 300 8 bf16 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/bench_train.py 16 300 3 bf16 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/train/kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = 4
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernel time per step %.2f ms, launches per step %.0f" % (tot / steps / 1e6, calls / steps))
for r in rows[:22]:
    print("%-90s calls/step %6.1f avg us %7.1f ms/step %6.2f" % (r["Name"][:90], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e6))
PY
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/train_kernel_stats.csv; rm -rf $O/kt
