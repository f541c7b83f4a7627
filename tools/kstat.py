"""average kernel durations of a rocprofv3 --kernel-trace --stats run:  python tools/kstat.py DIR [substring]"""
import csv, glob, sys
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            n = r["Name"].replace("void dimx::(anonymous namespace)::", "").replace("dimx::(anonymous namespace)::", "").split("(")[0]
            print("%-60s calls %5d avg %9.1f us" % (n[:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
