#!/bin/bash
# everything one forward of the headline workload launches outside the decode loop, in situ: GEMM shapes x kernels (gemm_in_situ.py) and
# the non-GEMM kernels of the prefill, from one rocprofv3 kernel trace of `bench.py --steps 1 --warmup 1`
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/insitu; rm -rf $O; mkdir -p $O
DIMX_GEMM_LOG=1 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-train-step --no-roofline > $O/line.json 2> $O/log.txt
python tools/gemm_in_situ.py $O/log.txt $O/kt > $O/gemm_in_situ.txt
python - <<'PY' > gpurun_out/insitu/other_kernels.txt
import csv, glob, collections
f = glob.glob("gpurun_out/insitu/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("dimx::", "").split("(")[0][:70]
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-72s n=%5d avg %8.1f us total %9.1f us (2 forwards)" % (k, len(v), sum(v) / len(v), sum(v)))
PY
rm -rf $O/kt
