#!/usr/bin/env python3
"""Merge the per-pass PMC summaries (tools/rocpd_pmc_summary.py) into one markdown table + JSON.
usage: pmc_table.py <dir with FETCH_SIZE.json WRITE_SIZE.json SQ.json SQ2.json> <out.md> <out.json> [header]"""
import json
import os
import sys

d, out_md, out_json = sys.argv[1:4]
hdr = sys.argv[4] if len(sys.argv) > 4 else ""
P = {n: json.load(open(os.path.join(d, n + ".json"))) for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2")}
F, W, S, S2 = P["FETCH_SIZE"], P["WRITE_SIZE"], P["SQ"], P["SQ2"]
lines = [hdr, "",
         "HBM bytes: FETCH_SIZE/WRITE_SIZE are KiB; per MI355X_MICROARCH.md FETCH_SIZE on gfx950 counts a 128-B request as 64 B for wide coalesced",
         "streams, so `read MB (x2)` doubles it (uncalibrated for the narrow/random accesses of the join kernels).  SQ_* cycle counters are",
         "quad-cycles summed over waves; `VALU busy %` = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, `wait any %` = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
         "(wave parked on s_waitcnt/barrier), `issue stall %` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.",
         "`waves/SIMD` = achieved occupancy: resident waves averaged over the kernel's busy time = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 8 (SQ_WAVE_CYCLES is in",
         "quad-cycles, SQ_BUSY_CYCLES is summed over the 32 shader engines, 1024 SIMDs: 4 x 32 / 1024; calibrated on kernels of known occupancy, profiles/r02_valu_rates_pmc.md).",
         "`LDS GB/s` = SQ_INSTS_LDS x 64 lanes x 4 B / kernel time (a lower bound: wider DS operations move more) against the 78.6 TB/s LDS peak (128 B/clk/CU);",
         "`LDS active %` = SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES; `bank conflict %` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.", "",
         "| kernel | disp | ms | read MB (x2) | write MB | waves | waves/SIMD | VALU inst/wave | LDS inst/wave | VALU busy % | issue stall % | wait any % | LDS GB/s (% of peak) | LDS active % | bank conflict % | LDS B/WG | VGPR |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(F, key=lambda k: -F[k]["total_ns"]):
    f = F[k]["counters"].get("FETCH_SIZE", 0); w = W.get(k, {}).get("counters", {}).get("WRITE_SIZE", 0)
    s = S.get(k, {}).get("counters", {}); s2 = S2.get(k, {}).get("counters", {})
    waves = s.get("SQ_WAVES", 0) or 1; wc = s.get("SQ_WAVE_CYCLES", 0) or 1; wc2 = s2.get("SQ_WAVE_CYCLES", 0) or 1
    secs = (S.get(k, F[k])["total_ns"] or 1) / 1e9
    lds_gbs = s.get("SQ_INSTS_LDS", 0) * 256 / secs / 1e9
    lines.append("| %s | %d | %.3f | %.1f | %.1f | %d | %.1f | %.0f | %.0f | %.1f | %.1f | %.1f | %.0f (%.1f) | %.1f | %.1f | %s | %s |" % (
        k.replace("skh::", "").replace("void ", ""), F[k]["dispatches"], F[k]["total_ns"] / 1e6, 2 * f / 1024, w / 1024, waves,
        s.get("SQ_WAVE_CYCLES", 0) / max(s.get("SQ_BUSY_CYCLES", 0), 1) / 8.0, s.get("SQ_INSTS_VALU", 0) / waves,
        s.get("SQ_INSTS_LDS", 0) / waves, 100 * s.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * s.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * s2.get("SQ_WAIT_ANY", 0) / wc2,
        lds_gbs, 100 * lds_gbs / 78643.0, 100 * s2.get("SQ_ACTIVE_INST_LDS", 0) / wc2, 100 * s.get("SQ_LDS_BANK_CONFLICT", 0) / max(s2.get("SQ_LDS_IDX_ACTIVE", 0), 1),
        F[k]["lds_bytes"], F[k]["vgprs"]))
open(out_md, "w").write("\n".join(lines) + "\n")
json.dump(P, open(out_json, "w"), indent=1, sort_keys=True)
print("\n".join(lines[7:22]))
