"""Per-kernel average of one PMC counter from a rocprofv3 --pmc ... --output-format csv run."""
import csv
import glob
import sys
from collections import defaultdict


def main(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        print(k)
        for c, v in cs.items():
            print("    %-14s n=%5d avg=%.1f" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
