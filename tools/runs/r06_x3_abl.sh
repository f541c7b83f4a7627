#!/bin/bash
# per-shape kernel durations of the split-bf16 decode GEMM (rocprofv3 kernel trace of tools/r06_x3_gemm_bench.py): as built, with 5 ring
# slots, and the two ablations (no LDS-DMA in the loop / no split + MFMA)
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
    if 'gemm_x3_kernel' in n or 'gemm_ws_kernel<float' in n:
        k=n[n.index('gemm_'):].split('(')[0]
        d[(k, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    v=sorted(v)[len(v)//10:]
    print("  %-44s %4d blocks  avg %7.2f us  min %7.2f  (%d launches)" % (k[0][:44],k[1],sum(v)/len(v)/1e3,min(v)/1e3,len(v)))
PY
}
: > $O/x3_gemm_abl.txt
for cfg in "" "DIMX_X3_STAGES=5" "DIMX_X3_ABL=1" "DIMX_X3_ABL=2" "DIMX_X3_ABL=3"; do
  echo "== ${cfg:-as built (4 ring slots)}" >> $O/x3_gemm_abl.txt
  rm -rf $O/ktx
  env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ktx -- python tools/r06_x3_gemm_bench.py > /dev/null 2>&1
  summ $O/ktx >> $O/x3_gemm_abl.txt
done
rm -rf $O/ktx
cat $O/x3_gemm_abl.txt
