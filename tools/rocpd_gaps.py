#!/usr/bin/env python3
"""Idle time between consecutive kernels of the last bench step in a rocprofv3 (rocpd sqlite) kernel trace.
usage: rocpd_gaps.py <results.db> [out.txt]   -- the step is taken from the last skh::seed_tiles_kernel dispatch to the end of the trace"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if "seed_tiles_kernel" in r[0]]
    i0 = starts[-1]
    # a large collection is seeded in several launches: the step starts at the first of them, i.e. behind the previous step's last estimate kernel
    prev = [i for i, r in enumerate(rows[:i0]) if "finalize_kernel" in r[0]]
    i0 = next(i for i in starts if i > (prev[-1] if prev else -1))
    step = rows[i0:]
    # cut at the last skh:: kernel (the torch kernels of the harness follow)
    last = max(i for i, r in enumerate(step) if "skh::" in r[0])
    step = step[:last + 1]
    busy = sum(r[2] - r[1] for r in step); span = step[-1][2] - step[0][1]
    out = ["step span %.3f ms, kernels %.3f ms, idle %.3f ms over %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(step))]
    gaps = []
    for a, b in zip(step, step[1:]):
        gaps.append((b[1] - a[2], a[0][:48], b[0][:48]))
    gaps.sort(reverse=True)
    out.append("largest gaps (us): after kernel -> before kernel")
    for g in gaps[:25]:
        out.append("%8.1f  %s -> %s" % (g[0] / 1e3, g[1], g[2]))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
