"""Summarise a rocprofv3 --kernel-trace results.db: per-kernel count / total / average, like --stats."""
import sqlite3
import sys


def main(path, top=25):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    print("%7s %7s %10s %9s %9s %9s  %s" % ("share", "calls", "total_ms", "avg_us", "min_us", "max_us", "kernel"))
    for r in rows[:top]:
        print("%6.2f%% %7d %10.3f %9.1f %9.1f %9.1f  %s" % (100 * r[2] / tot, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3,
                                                          r[5] / 1e3, r[0][:120]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
