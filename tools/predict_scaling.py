#!/usr/bin/env python3
"""Inputs for DESIGN.md's PREDICTED 1/2/4/8-GPU series of the strong-scaling mode (bench.py --collection 10000), measured on ONE uncontended MI355X:
the phases that shard linearly (seeding, marker sets, seed tables, chaining) are timed once on the whole collection; the screen, which does not, is timed
per world size W through the public key-range entry points (skh_screen_part for every part of W, the slowest counted + skh_screen_from_cells over the cells of all parts = what
one rank of a world of W runs).  Prints one JSON object; nothing here is a multi-GPU measurement.
usage: predict_scaling.py [genomes=10000]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import skani_amd as sk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
ctx = sk.Context(0)
canon = bench.genome_order(n, "shuffled")
bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, np.sort(canon))       # (generated clade by clade; the order does not matter for the phase times)
torch.cuda.synchronize()
gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
del bases
torch.cuda.empty_cache()
params = sk.SketchParams(125, 15, 1000, sk.SEED_AVX2)
mp = sk.MapParams(learned_ani=True, compute_ci=True)
out = {"genomes": n, "bases": int(coff[-1])}
for rep in range(2):                                                           # second pass reported
    ctx.timings()
    ss = ctx.sketch_genomes(gs, params, genome_rank=np.arange(n, dtype=np.uint32))
    i, j, res, nch = ctx.triangle(ss, mp)
    tm = ctx.timings()
    if rep == 0:
        ss.close()
out["one_gpu_ms"] = {k: tm[k] for k in ("seed_ms", "sketch_build_ms", "screen_ms", "chain_ms")}
out["chained_pairs"] = int(nch)
scr = {}
for W in (1, 2, 4, 8):
    cells = [ctx.screen_part(ss, p, W) for p in range(W)]
    allc = np.concatenate(cells)
    part_ms = []
    for p in range(W):                                                          # every part timed: the slowest rank is what a step waits for
        b = None
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.screen_part(ss, p, W)
            torch.cuda.synchronize(); b = min(b or 1e9, (time.perf_counter() - t0) * 1e3)
        part_ms.append(b)
    # the cells of all parts through the rule: in row order (what the distributed call's gathered blocks are: added up row by row in LDS, round 5) and as concatenated
    # (out of row order: the dense N x N matrix, the form of rounds 3-4).  (The host copy of the cells up to the device is inside both: the distributed call has them there.)
    fc = {}
    for form, arr in (("rows", np.sort(allc)), ("dense", allc)):
        best = None
        for rep in range(3):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            a, b = ctx.screen_from_cells(ss, arr, 0.0, True)
            torch.cuda.synchronize(); best = min(best or 1e9, (time.perf_counter() - t1) * 1e3)
        assert len(a) == nch, (len(a), nch)
        fc[form] = best
    scr[str(W)] = {"part_ms_max": max(part_ms), "part_ms": [round(x, 3) for x in part_ms], "from_cells_ms": fc["rows"], "from_cells_dense_ms": fc["dense"],
                   "cells_per_part": [int(len(c)) for c in cells], "cells_total": int(len(allc))}
out["screen_by_key_range_ms"] = scr
ctx.timings()
a, b = ctx.screen(ss, None, 0.0, 0, True)
out["screen_full_ms"] = ctx.timings()["screen_ms"]
print(json.dumps(out))
