"""Kernel timeline of the LAST step in a rocprofv3 results .db (kernel-trace): one line per launch with
its duration and the idle gap since the previous kernel ended; the sum of the gaps is what launch
latency and stream dependencies cost on top of the kernels themselves.

    python tools/rocprof_timeline.py results.db [first-kernel-substring]
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_analyze"
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    # a step starts at a `first` kernel that follows a non-`first` kernel
    starts = [i for i, r in enumerate(rows) if first in r[0] and (i == 0 or first not in rows[i - 1][0])]
    if len(starts) < 2:
        print("no complete step found")
        return
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t_prev = None
    busy = gaps = 0.0
    for name, s, e in step:
        gap = 0.0 if t_prev is None else (s - t_prev) / 1e3
        print("%-60s %9.1f us   gap %7.1f us" % (name[:60], (e - s) / 1e3, gap))
        busy += (e - s) / 1e3
        gaps += max(gap, 0.0)
        t_prev = e
    print("kernels %.1f us + gaps %.1f us = %.1f us from first start to last end (%d launches)" %
          (busy, gaps, (step[-1][2] - step[0][1]) / 1e3, len(step)))


if __name__ == "__main__":
    main()
