#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
summ() { python - "$1" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'gemm_x3_kernel' in n:
        k=n[n.index('gemm_'):].split('(')[0]
        d[(k, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    v=sorted(v)[len(v)//10:]
    print("  %-30s %4d blocks  avg %7.2f us  min %7.2f" % (k[0][:30],k[1],sum(v)/len(v)/1e3,min(v)/1e3))
PY
}
: > $O/x3_gemm_abl2.txt
for cfg in "" "DIMX_X3_ABL=8" "DIMX_X3_ABL=4" "DIMX_X3_ABL=12" "DIMX_X3_ABL=9"; do
  echo "== ${cfg:-as built}" >> $O/x3_gemm_abl2.txt
  rm -rf $O/ktx
  env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ktx -- python tools/r06_x3_gemm_bench.py > /dev/null 2>&1
  summ $O/ktx >> $O/x3_gemm_abl2.txt
done
rm -rf $O/ktx
cat $O/x3_gemm_abl2.txt
