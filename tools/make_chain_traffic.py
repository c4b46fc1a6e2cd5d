#!/usr/bin/env python3
"""Writes profiles/chain_traffic.json: the HBM traffic of the chaining stage per bench step as ONE number (what bench.py quotes as roofline_chain.traffic).
usage: make_chain_traffic.py <pmc json from tools/pmc.sh (gpurun_out/pmc_<tag>.json)> <bench json of the same code> <source note>

FETCH_SIZE on gfx950 counts a coalesced stream's 128-byte requests as 64 B and a random access to a 64-byte line as 64 B (profiles/r02_fetch_calib.md), so a
kernel's read bytes are  2 x counted  for streams and  1 x counted  for scattered accesses; WRITE_SIZE is exact for streams.  Every kernel of the chaining
stage is put in one class by how it reads:
  stream   join_fill (hit records in order), chunk_stats (256-byte position blocks), the scans, the order kernels, finalize
  random   chunk_kernel (binary-search probes), chain_dp_* (every lane walks its own chunk, 32 bytes at a time), greedy_* (interval records by index), slot tables
  mixed    join_count: S bytes of hashes / positions streamed (8 B per enumerated position, known from the workload) + random table probes:
           counted = S / 2 + R  =>  read = counted + S / 2
The file carries a hash of the chaining sources; bench.py refuses it when they changed."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(sys.argv[1])); bench = json.load(open(sys.argv[2])); note = sys.argv[3] if len(sys.argv) > 3 else ""
CHAIN = ("slot_tile", "join_count", "join_fill", "chunk_kernel", "dp_order", "chain_dp", "interval_emit", "greedy", "chunk_stats", "finalize")
STREAM = ("join_fill", "chunk_stats", "dp_order", "finalize", "greedy_order")
cfg = bench["config"]
positions_per_genome = cfg["bases_per_gpu"] / cfg["genomes_per_gpu"] / 125.0 if "-c 125" in cfg["workload"] else None
S = 8.0 * cfg["chained_pairs"] * (positions_per_genome or 0)                     # bytes of hashes + positions the count pass streams per step
rows, total_r, total_w = [], 0.0, 0.0
for k, f in pmc["FETCH_SIZE"].items():
    name = k.replace("skh::", "").replace("void ", "")
    if not any(c in name for c in CHAIN):
        continue
    d = f["dispatches"]
    counted = f["counters"]["FETCH_SIZE"] * 1024 / d
    w = pmc["WRITE_SIZE"].get(k, {}).get("counters", {}).get("WRITE_SIZE", 0) * 1024 / d
    if "join_count" in name:
        cls, read = "mixed", counted + S / 2
    elif any(c in name for c in STREAM):
        cls, read = "stream", 2 * counted
    else:
        cls, read = "random", counted
    rows.append({"kernel": name[:60], "class": cls, "launches_per_step": d, "read_bytes": read * d, "write_bytes": w * d, "ms": f["total_ns"] / 1e6})
    total_r += read * d; total_w += w * d
alg = bench["roofline_chain"]["bytes_per_step"]
srcs = sorted(f for f in os.listdir(os.path.join(ROOT, "skani_amd", "csrc")) if f.startswith("chain"))
sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "skani_amd", "csrc", f), "rb").read() for f in srcs)).hexdigest()[:16]
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out = {"hbm_bytes_per_step": total_r + total_w, "read_bytes_per_step": total_r, "write_bytes_per_step": total_w, "algorithmic_bytes_per_step": alg,
       "traffic_over_algorithmic": (total_r + total_w) / alg, "streamed_by_count_pass": S, "kernels": sorted(rows, key=lambda r: -(r["read_bytes"] + r["write_bytes"])),
       "chain_sources_sha256_16": sha, "sources": srcs, "commit": commit + " (+ working tree)", "source": note,
       "rule": "read = 2 x FETCH_SIZE for streaming kernels, 1 x for scattered ones, FETCH_SIZE + S/2 for the count pass (profiles/r02_fetch_calib.md); one PMC pass = one bench step"}
json.dump(out, open(os.path.join(ROOT, "profiles", "chain_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
for r in out["kernels"]:
    print("%-60s %-7s %8.1f MB read %8.1f MB written %7.3f ms" % (r["kernel"], r["class"], r["read_bytes"] / 1e6, r["write_bytes"] / 1e6, r["ms"]))
