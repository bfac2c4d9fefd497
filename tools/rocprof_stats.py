"""Summarise a rocprofv3 results .db (kernel-trace) as a per-kernel table: calls, total/avg/min/max us, share."""
import sqlite3
import sys


def summarise(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    out = ["%-72s %6s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%")]
    for r in rows:
        out.append("%-72s %6d %12.1f %10.1f %10.1f %10.1f %6.1f" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / total))
    return "\n".join(out)


if __name__ == "__main__":
    print(summarise(sys.argv[1]))
