#!/usr/bin/env python3
"""Timeline of the last bench step in a rocprofv3 (rocpd sqlite) kernel trace: start / end (us, relative to the step's first kernel) and queue of every kernel.
usage: rocpd_timeline.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    gcol = next((c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols), None)
    wcol = next((c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols), None)
    rows = list(db.execute("select name, start, end, %s, %s, %s from kernels order by start" % (qcol or "0", gcol or "0", wcol or "0")))
    i0 = [i for i, r in enumerate(rows) if "seed_tiles_kernel" in r[0]][-1]
    # a large collection is seeded in several launches: the step starts at the first of them, i.e. behind the previous step's last estimate kernel
    prev = [i for i, r in enumerate(rows[:i0]) if "finalize_kernel" in r[0]]
    i0 = next(i for i in range(prev[-1] + 1 if prev else 0, i0 + 1) if "seed_tiles_kernel" in rows[i][0])
    step = rows[i0:]
    last = max(i for i, r in enumerate(step) if "skh::" in r[0])
    step = step[:last + 1]
    t0 = step[0][1]
    out = ["%9s %9s %8s %s %10s %5s  kernel" % ("start us", "end us", "dur us", qcol or "q", "grid", "wg")]
    for r in step:
        out.append("%9.1f %9.1f %8.1f %s %10s %5s  %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, str(r[3]), str(r[4]), str(r[5]), r[0].replace("void ", "")[:90]))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
