"""print per-kernel (calls, avg ns) of two rocprofv3 --stats runs side by side: python tools/ab_profile.py A.csv B.csv"""
import csv
import re
import sys


def load(p):
    out = {}
    for r in csv.DictReader(open(p)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\(.*", "", name)[:90]
        out[name] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0, 0))[2] + b.get(k, (0, 0, 0))[2]))
for k in keys[:14]:
    ca, aa, ta = a.get(k, (0, 0, 0))
    cb, ab, tb = b.get(k, (0, 0, 0))
    print("%-92s %6d %9.0f | %6d %9.0f  %+5.1f%%" % (k, ca, aa, cb, ab, 100 * (ab / aa - 1) if aa and ab else 0))
