#!/usr/bin/env python3
"""Writes profiles/stage_profile.json: every kernel of a bench step with its stage, its average duration in a plain `rocprofv3 --kernel-trace` run, its HBM bytes from
the FETCH_SIZE / WRITE_SIZE passes and its residency from the SQ passes -- what bench.py puts on its line as `roofline_stages` (and as the rocprof time of `roofline.frac`).
usage: make_stage_profile.py <trace json (tools/prof.sh: gpurun_out/trace_<tag>.json)> <pmc json (tools/pmc.sh: gpurun_out/pmc_<tag>.json)> <steps in the trace> <source note>

Read bytes follow profiles/r02_fetch_calib.md: FETCH_SIZE counts a coalesced stream's 128-byte requests as 64 B (x 2 for streaming kernels) and a scattered access to a
64-byte line as 64 B (x 1); the join's count pass is a stream (8 B per enumerated position) plus scattered probes: the stream's half is added when the line is made, where
the number of enumerated positions is known (bench.py).  The file carries a hash of ALL kernel sources; bench.py refuses it when they changed."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
trace = json.load(open(sys.argv[1])); pmc = json.load(open(sys.argv[2])); steps = int(sys.argv[3]); note = sys.argv[4] if len(sys.argv) > 4 else ""
STAGES = (  # first match wins
    ("ingest (outside the step)", ("pack_kernel", "nmask")),
    ("seeding", ("seed_tiles", "seed_offsets", "seed_got", "seed_compact")),
    ("tables", ("slice_positions", "table_blocks", "build_tables", "marker_set", "marker_gather", "fill_regions", "unpack_positions", "pack_positions")),
    ("screen index + column order (at sketch time, second stream)", ("skeys_", "colorder_", "screen_count_tri_rows_kernel<false, 256u>")),
    ("screen", ("screen_",)),
    ("join", ("expand_pairs", "slot_tile", "join_count", "join_fill", "widen_anchors")),
    ("chunking + DP", ("chunk_kernel", "dp_order", "chain_dp", "interval_emit")),
    ("selection + estimate", ("greedy", "chunk_stats", "finalize")),
)
STREAM = ("seed_tiles", "seed_compact", "pack_kernel", "slice_positions", "build_tables", "marker_", "skeys_", "screen_threshold", "join_fill", "chunk_stats", "dp_order", "finalize", "greedy_order", "scan_")
def short(k): return k.replace("void ", "").replace("skh::", "")
rows = []
for k, t in trace.items():
    if "skh::" not in k: continue
    name = short(k)
    stage = next((s for s, pats in STAGES if any(p in name for p in pats)), "scans and fills")
    f = pmc["FETCH_SIZE"].get(k); w = pmc["WRITE_SIZE"].get(k); s = pmc["SQ"].get(k, {}).get("counters", {}); s2 = pmc["SQ2"].get(k, {}).get("counters", {})
    cls = "mixed" if "join_count" in name else ("stream" if any(p in name for p in STREAM) else "random")
    row = {"kernel": name, "stage": stage, "class": cls, "launches_per_step": t["calls"] / steps, "ms_per_step": t["total_ms"] / steps, "avg_us": t["avg_us"],
           "vgprs": t.get("vgprs"), "lds_bytes": t.get("lds_bytes")}
    if f:
        counted = f["counters"]["FETCH_SIZE"] * 1024                                  # one PMC pass = one bench step
        row["read_bytes_per_step"] = counted * (2 if cls == "stream" else 1)
        row["write_bytes_per_step"] = (w or {}).get("counters", {}).get("WRITE_SIZE", 0) * 1024
    if s.get("SQ_WAVES"):
        wc = s.get("SQ_WAVE_CYCLES", 0) or 1
        row["waves_per_step"] = s["SQ_WAVES"]; row["waves_per_simd"] = s.get("SQ_WAVE_CYCLES", 0) / max(s.get("SQ_BUSY_CYCLES", 0), 1) / 8.0
        row["valu_busy_pct"] = 100.0 * s.get("SQ_ACTIVE_INST_VALU", 0) / wc
        if s2.get("SQ_WAVE_CYCLES"): row["wait_any_pct"] = 100.0 * s2.get("SQ_WAIT_ANY", 0) / s2["SQ_WAVE_CYCLES"]
    rows.append(row)
srcdir = os.path.join(ROOT, "skani_amd", "csrc")
srcs = sorted(f for f in os.listdir(srcdir) if f.endswith((".hip", ".h")))
sha = hashlib.sha256(b"".join(open(os.path.join(srcdir, f), "rb").read() for f in srcs)).hexdigest()[:16]
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out = {"kernel_sources_sha256_16": sha, "commit": commit + " (+ working tree)", "source": note, "steps_in_trace": steps,
       "workload": "bench.py default: 1000 genomes x 5 Mbp, clade order, -c 125", "kernels": sorted(rows, key=lambda r: -r["ms_per_step"]),
       "rule": "ms: plain rocprofv3 --kernel-trace averages; read = 2 x FETCH_SIZE for streaming kernels, 1 x for scattered ones (profiles/r02_fetch_calib.md), the count pass's stream added by bench.py; "
               "waves_per_simd = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 8 (profiles/r02_valu_rates_pmc.md)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "stage_profile.json"), "w"), indent=1)
by = {}
for r in rows: by.setdefault(r["stage"], 0.0); by[r["stage"]] += r["ms_per_step"]
print(json.dumps({k: round(v, 3) for k, v in by.items()}, indent=1))
