#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_gemm_x3.py -x -q 2>&1 | tail -4 || exit 1
if ! timeout 300 python -m pytest tests/test_gpu_gemm_x3.py -x -q > /dev/null 2>&1; then echo 'x3 tests FAILED: stopping'; exit 1; fi
B="timeout 300 python bench.py --mode f32 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-train-step"
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %.1f clips/s %.2f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
{
DIMX_NO_X3=1 $B 2>/dev/null | pr "f32 mode, f32 MFMA decode GEMMs (DIMX_NO_X3=1)"
$B 2>/dev/null | pr "f32 mode, split-bf16 decode GEMMs"
} | tee $O/x3_ab2.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32 -- python bench.py --mode f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step --no-mode-compare > /dev/null 2>&1
cp $(ls $O/kt32/*/*kernel_stats.csv | head -1) $O/x3_parity_kernel_stats.csv
python - <<'PY' | tee gpurun_out/r06/x3_by_grid.txt
import csv,glob
f=glob.glob('gpurun_out/r06/kt32/*/*kernel_trace.csv')[0]
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'gemm_x3' in n:
        d[(n[n.index('gemm_x3'):].split('(')[0], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    print("%-28s %4d blocks x %6d launches  avg %7.2f us  min %7.2f" % (k[0],k[1],len(v),sum(v)/len(v)/1e3,min(v)/1e3))
PY
rm -rf $O/kt32
