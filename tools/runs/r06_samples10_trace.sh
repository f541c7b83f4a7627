#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/kts
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- python bench.py --steps 2 --warmup 1 --samples 10 --no-cpu-baseline --no-parity-mode --no-roofline --no-train-step > $O/samples10_under_rocprof.json 2>/dev/null
cp $(ls $O/kts/*/*kernel_stats.csv | head -1) $O/samples10_kernel_stats.csv
rm -rf $O/kts
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06/samples10_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6)
for r in rows[:18]:
    print("%-95s %6d %9.2f ms %8.2f us" % (r['Name'][:95], int(r['Calls']), int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
