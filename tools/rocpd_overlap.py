#!/usr/bin/env python3
"""How much of the skh:: kernel time overlaps between streams (rocpd sqlite kernel trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels where name like '%skh::%' order by start"))
tot = sum(e - s for _, s, e, _, _ in rows)
# union length
iv = sorted((s, e) for _, s, e, _, _ in rows)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print("kernels", len(rows), "sum ms %.2f union ms %.2f overlap ms %.2f" % (tot / 1e6, u / 1e6, (tot - u) / 1e6), "streams", sorted(set(r[3] for r in rows)), "queues", sorted(set(r[4] for r in rows)))
# print a slice of the timeline of the last step
last = [r for r in rows if r[1] > rows[-1][2] - 30e6]
t0 = last[0][1]
for n, s, e, st, q in last[:60]:
    print("%8.3f %8.3f  s%s q%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, st, q, n.split("(")[0][-40:]))
