#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/kts
DIMX_GEMM_LOG=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kts -- python bench.py --steps 1 --warmup 1 --samples 10 --no-cpu-baseline --no-parity-mode --no-roofline --no-train-step > /dev/null 2> $O/s10_gemm_log.txt
python - <<'PY'
import csv,glob
from collections import defaultdict
f=glob.glob('gpurun_out/r06/kts/*/*kernel_trace.csv')[0]
d=defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'gemm' in n:
        k=n[n.index('gemm'):].split('(')[0][:60]
        d[(k,int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if len(v)<200: continue
    print("%-62s %5d blocks x %6d  avg %7.2f us  total %7.1f ms" % (k[0],k[1],len(v),sum(v)/len(v)/1e3,sum(v)/1e6))
PY
sort $O/s10_gemm_log.txt | grep "M=2560" | uniq -c | sort -rn | head -12
rm -rf $O/kts
