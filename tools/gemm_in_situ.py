"""Which GEMM shapes one forward of the headline workload launches outside the decode loop, and what each costs IN SITU:
pairs the DIMX_GEMM_LOG=1 lines of a run with the GEMM rows of its rocprofv3 kernel trace (same order).
    DIMX_GEMM_LOG=1 DIMX_NO_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 1 --warmup 1 ... 2> LOG
    python tools/gemm_in_situ.py LOG DIR"""
import collections
import csv
import glob
import sys

log, d = sys.argv[1], sys.argv[2]
shapes = [l.strip()[len("dimx-gemm "):] for l in open(log, errors="replace") if l.startswith("dimx-gemm ")]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gemms = [r for r in rows if "gemm" in r["Kernel_Name"] and "fused_probe" not in r["Kernel_Name"]]
print("log lines %d, gemm kernels in the trace %d" % (len(shapes), len(gemms)))
agg = collections.OrderedDict()
n = min(len(shapes), len(gemms))
for sh, r in zip(shapes[:n], gemms[:n]):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("dimx::", "").split("(")[0][:60]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault((sh, k), []).append(us)
tot = 0.0
for (sh, k), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if "M=256 " in sh or "M=128 " in sh:
        continue
    tot += sum(v)
    print("%-95s %-48s n=%4d avg %8.1f us  total %9.1f us" % (sh, k, len(v), sum(v) / len(v), sum(v)))
print("total (prefill-sized) %.1f us over the traced run" % tot)
