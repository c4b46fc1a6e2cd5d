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
        if sys.argv[2].endswith(".md"):                                              # the same rows for tools (tools/make_stage_profile.py), names as tools/rocpd_pmc_summary.py writes them
            import json
            js = {}
            for r in rows:
                k = r[0].replace("(anonymous namespace)::", "").split("(")[0]
                e = js.setdefault(k, {"calls": 0, "total_ms": 0.0, "vgprs": r[6], "lds_bytes": r[7], "scratch_bytes": r[8]})
                e["calls"] += r[1]; e["total_ms"] += r[2]
            for e in js.values(): e["avg_us"] = 1e3 * e["total_ms"] / max(e["calls"], 1)
            json.dump(js, open(sys.argv[2][:-3] + ".json", "w"), indent=1, sort_keys=True)
    else:
        print(text)


if __name__ == "__main__":
    main()
