#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max.
usage: rocpd_summary.py <results.db> [out.md] [header line]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                           "max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | lds B | scratch B |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s |" % (r[0][:90].replace("|", "/"), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8]))
    hdr = (sys.argv[3] + "\n\n") if len(sys.argv) > 3 else ""
    text = hdr + "total kernel time %.3f ms\n\n" % tot + "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
