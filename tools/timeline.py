"""Round 6: what runs beside what?  Reads a rocprofv3 kernel trace (`*_kernel_trace.csv`, --kernel-trace --output-format csv) and
reports, for the busiest window of decode-step kernels:
  * per queue: launches, busy time, share of the window;
  * how much of the window has 0 / 1 / >= 2 kernels in flight (all queues together);
  * "HBM stream" time: the union of the intervals of the decode attention / layer kernels (the HBM-bound launches), and how much of
    it has such kernels of TWO queues in flight at once (two engines streaming against each other);
  * average duration per kernel name, per queue (a kernel that takes longer beside another engine shows here).

usage: python tools/timeline.py DIR_OR_CSV [skip_fraction]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

STREAM_KEYS = ("decode_attn", "xcd_layer", "attn_clip")


def find_csv(p):
    if os.path.isfile(p):
        return p
    c = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))
    if not c:
        raise SystemExit("no *kernel_trace.csv under " + p)
    return c[0]


def union_len(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def depth_hist(iv):
    """time with k intervals in flight"""
    ev = []
    for s, e in iv:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = defaultdict(int)
    d, last = 0, None
    for t, k in ev:
        if last is not None and d > 0:
            hist[d] += t - last
        d += k
        last = t
    return hist


def main():
    path = find_csv(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = list(csv.DictReader(open(path)))
    if not rows:
        raise SystemExit("empty trace")
    k = rows[0].keys()
    qk = "Queue_Id" if "Queue_Id" in k else ("Stream_Id" if "Stream_Id" in k else None)
    ev = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, e, r.get(qk, "0") if qk else "0", r["Kernel_Name"].split("(")[0][:70]))
    ev.sort()
    # window: the last (1 - skip) of the trace (warm-up, prefill of the first step and the roofline legs come first)
    t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
    w0 = t0 + int((t1 - t0) * skip)
    ev = [x for x in ev if x[0] >= w0]
    span = max(e for _, e, _, _ in ev) - ev[0][0]
    print("trace %s\nwindow: last %.0f %% of the trace = %.2f ms, %d launches" % (path, 100 * (1 - skip), span / 1e6, len(ev)))
    byq = defaultdict(list)
    for s, e, q, n in ev:
        byq[q].append((s, e))
    for q, iv in sorted(byq.items(), key=lambda x: -len(x[1])):
        if len(iv) < 50:
            continue
        print("  queue %-6s %6d launches, busy %8.2f ms = %5.1f %% of the window" % (q, len(iv), union_len(iv) / 1e6, 100.0 * union_len(iv) / span))
    allv = [(s, e) for s, e, _, _ in ev]
    h = depth_hist(allv)
    busy = sum(h.values())
    print("kernels in flight: 0 -> %.1f %%, 1 -> %.1f %%, 2 -> %.1f %%, >= 3 -> %.1f %% of the window" % (
        100.0 * (span - busy) / span, 100.0 * h.get(1, 0) / span, 100.0 * h.get(2, 0) / span,
        100.0 * sum(v for d, v in h.items() if d >= 3) / span))
    st = [(s, e, q) for s, e, q, n in ev if any(key in n for key in STREAM_KEYS)]
    if st:
        u = union_len([(s, e) for s, e, _ in st])
        hq = depth_hist([(s, e) for s, e, _ in st])
        two = sum(v for d, v in hq.items() if d >= 2)
        print("HBM-stream kernels (%s): in flight %.1f %% of the window; two or more of them at once %.1f %% of the window" % (
            " / ".join(STREAM_KEYS), 100.0 * u / span, 100.0 * two / span))
    names = defaultdict(lambda: defaultdict(list))
    for s, e, q, n in ev:
        names[n][q].append(e - s)
    print("average duration per kernel (us), by queue:")
    tot = {n: sum(sum(v) for v in d.values()) for n, d in names.items()}
    for n in sorted(tot, key=lambda x: -tot[x])[:14]:
        parts = ["q%s: %7.2f x %5d" % (q, sum(v) / len(v) / 1e3, len(v)) for q, v in sorted(names[n].items()) if len(v) >= 10]
        print("  %-62s %s" % (n[:62], "   ".join(parts)))


if __name__ == "__main__":
    main()
