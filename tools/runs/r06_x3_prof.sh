#!/bin/bash
O=gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32 -- python bench.py --mode f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step > /dev/null 2>&1
cp $(ls $O/kt32/*/*kernel_stats.csv | head -1) $O/x3_parity_kernel_stats.csv
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r06/kt32/*/*kernel_trace.csv')[0]
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'gemm_x3' in n or 'gemm_ws_kernel<float' in n:
        d[(n.split('(')[0][-40:], r['Grid_Size_X'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    print("%-44s grid %-7s x %6d  avg %7.2f us  min %7.2f" % (k[0],k[1],len(v),sum(v)/len(v)/1e3,min(v)/1e3))
PY
head -12 $O/x3_parity_kernel_stats.csv | cut -c1-160
rm -rf $O/kt32
bash tools/r06_layer_weights_abl.sh 2>/dev/null | tee $O/layer_weights_abl.txt
