#!/usr/bin/env python3
"""the PREDICTED strong-scaling series (bench.py --collection 10000 on 1 / 2 / 4 / 8 GPUs) as a markdown table.  Nothing here is a multi-GPU
measurement: the inputs are one-GPU measurements (profiles/r04_predict_inputs.json from tools/predict_scaling.py, the N = 1 bench line, the per-rank byte counts of
the 8-rank run on one device) and the link figures of MI355X_MICROARCH.md.
usage: predict_table.py [profiles dir]"""
import json
import os
import sys

P = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
R = os.environ.get("PREDICT_ROUND", "r06")
inp = json.load(open(os.path.join(P, R + "_predict_inputs.json")))
n1 = json.loads(open(os.path.join(P, R + "_bench_config4_n1.json")).read().strip().splitlines()[-1])
ph = n1["phase_ms_per_step"]
N = inp["genomes"]
LINK = 76.5e9 * 0.6            # one xGMI link, one direction: 153 GB/s bidirectional, 60 % of it assumed reachable by RCCL
MARKER_BYTES = 5000 * 8 * N   # ~5,000 markers per 5 Mbp genome at marker_c = 1000
MARKERS_MS, TABLES_MS = 2.3, ph["sketch_build_ms"] - 2.3       # the sketch phase's two parts at N = 10,000 (SKH_TRACE: markers 2.3 ms beside the tables)
# measured at 10,000 genomes with a world of one (SKH_TRACE=1 marks): the plan 0.60 ms; the result rows -- gathered on rank 0 only since round 5 (SKH_DIST_ROWS_TO_ROOT: the
# others return their own) -- 0.7 ms on rank 0 (95,000 rows of 72 B placed by one walk), which the step waits for (round 4: 2.4 ms on every rank)
PLAN_MS, RESULTS_MS = 0.6, 0.7
HOST_MS = n1["ms_per_step"] - sum(ph.values())                  # what the N = 1 step spends outside the library's phase timers (Python, result copies, waits)
# sketches a rank receives, GB: the plan (skh_plan_pairs, host code) run on config 4's shape for every world size -- 2,102 / ~1,600 / ~950 genomes of 39,600
# positions x 8 B; 8 ranks on one device measured 0.298-0.313 GB
recv_gb = {1: 0.0, 2: 0.67, 4: 0.51, 8: 0.31}
rows = []
for W in (1, 2, 4, 8):
    scr = inp["screen_by_key_range_ms"][str(W)]
    seed = ph["seed_ms"] / W
    markers = MARKERS_MS / W
    # round 5: the marker sets travel by KEY RANGE -- a rank receives its own W-th of every peer's sets (an all-to-all): MARKER_BYTES / W / W from each peer over that
    # peer's own link, all links at once (round 4's all-gather moved W times as much to every rank)
    ag = 0.0 if W == 1 else MARKER_BYTES / W / W / LINK * 1e3
    screen = ph["screen_ms"] if W == 1 else scr["part_ms_max"] + scr["from_cells_ms"]
    tables = TABLES_MS / W
    xfer = 0.0 if W == 1 else recv_gb[W] * 1e9 / (W - 1) / LINK * 1e3
    exposed = max(0.0, xfer - tables)                                                # the exchange runs beside the home set's table build
    chain = ph["chain_ms"] / W
    fixed = (0.0 if W == 1 else PLAN_MS + RESULTS_MS) + HOST_MS
    total = seed + markers + ag + screen + tables + exposed + chain + fixed
    rows.append((W, seed, markers + ag, screen, tables, exposed, chain, fixed, total))
t1 = rows[0][-1]
print("| GPUs | seeding | marker sets + their exchange | screen (key range + cells) | seed tables | exchange not hidden | chaining | plan, results, host | step | speed-up | pairs/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | **%.1f ms** | %.2f | %.0f M |" % (r + (t1 / r[-1], N * (N - 1) / 2 / r[-1] / 1e3)))
print("\nmeasured N = 1: %.1f ms per step (%s)" % (n1["ms_per_step"], "profiles/%s_bench_config4_n1.json" % R))
