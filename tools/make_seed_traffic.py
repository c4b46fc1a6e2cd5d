#!/usr/bin/env python3
"""Writes profiles/seed_traffic.json: what bench.py quotes as roofline.traffic / roofline.valu for the seeding kernel.
usage: make_seed_traffic.py <pmc json from tools/pmc.sh (gpurun_out/pmc_<tag>.json)> <source note> [kernel-trace summary of the same code (tools/prof.sh: trace_<tag>.md)]
With the trace summary the file also carries rocprof_avg_ms, the kernel's AVERAGE duration in a plain `rocprofv3 --kernel-trace` run (no counters): bench.py quotes
roofline.frac_rocprof from it beside the fraction from its own HIP-event time.
HBM bytes per launch = FETCH_SIZE (KiB, doubled: a coalesced stream's 128-byte requests count as 64 B on gfx950 -- profiles/r02_fetch_calib.md) + WRITE_SIZE (KiB);
VALU issue cycles per wave = measured SQ_INSTS_VALU per wave x the mean cost of the kernel's static instruction mix (tools/isa_mix.py, rates of
profiles/r02_valu_rates.md).  The file carries the hash of skani_amd/csrc/pack_seed.hip it was measured on; bench.py refuses it when the source changed."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc = json.load(open(sys.argv[1])); note = sys.argv[2] if len(sys.argv) > 2 else ""
key = next(k for k in pmc["FETCH_SIZE"] if "seed_tiles_kernel" in k)
f = pmc["FETCH_SIZE"][key]; w = pmc["WRITE_SIZE"][key]; s = pmc["SQ"][key]
launches = f["dispatches"]
fetch_b = 2 * f["counters"]["FETCH_SIZE"] * 1024 / launches; write_b = w["counters"]["WRITE_SIZE"] * 1024 / launches
valu_per_wave = s["counters"]["SQ_INSTS_VALU"] / s["counters"]["SQ_WAVES"]
src = os.path.join(ROOT, "skani_amd", "csrc", "pack_seed.hip")
mix = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), src, "seed_tiles_kernelILb1", "--dynamic-valu", str(valu_per_wave)],
                                check=True, capture_output=True, text=True).stdout)
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out = {"kernel": key, "hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
       "kernel_source_sha256_16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16], "commit": commit + " (+ working tree)", "source": note,
       "waves_per_launch": s["counters"]["SQ_WAVES"] / launches,
       "valu": {"clock_ghz": 2.38, "valu_inst_per_wave": valu_per_wave, "mean_cycles_per_valu": mix["mean_cycles_per_valu"], "issue_cycles_per_wave": mix["issue_cycles_per_wave"],
                "static_by_class": mix["static_by_class"], "cycles_per_class": mix["cycles_per_class"],
                "note": "issue cycles one wave needs = measured VALU instructions per wave x mean cost of the kernel's static mix (full-rate 2.17 / half-rate 4.27 / 4.5 cycles, profiles/r02_valu_rates.md)"}}
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    for line in open(sys.argv[3]):
        if "seed_tiles_kernel<true>" in line and line.startswith("|"):
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            out["rocprof_avg_ms"] = float(cells[3]) / 1000.0; out["rocprof_min_ms"] = float(cells[4]) / 1000.0; out["rocprof_calls"] = int(cells[1])
            out["rocprof_source"] = os.path.basename(sys.argv[3])
            break
json.dump(out, open(os.path.join(ROOT, "profiles", "seed_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
