#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 rocpd sqlite file (counters_collection view).
usage: rocpd_pmc_summary.py <results.db> <out.json> [name-filter]   (only kernels whose name contains the filter)"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1]); flt = sys.argv[3] if len(sys.argv) > 3 else "skh::"
    rows = db.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration), max(lds_block_size), max(vgpr_count), max(workgroup_size), "
                      "sum(grid_size) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ("%" + flt + "%",))
    out = {}
    for name, counter, n, val, dur, lds, vgpr, wg, grid in rows:
        k = name.replace("(anonymous namespace)::", "").split("(")[0]                 # (the kernels of an unnamed namespace keep their names: they all became "skh::" before round 6)
        d = out.setdefault(k, {"dispatches": n, "total_ns": dur, "lds_bytes": lds, "vgprs": vgpr, "workgroup": wg, "threads": grid, "counters": {}})
        d["counters"][counter] = val
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v["counters"] for k, v in out.items()}, indent=0)[:3000])


if __name__ == "__main__":
    main()
