#!/usr/bin/env python3
"""Inputs for DESIGN.md's PREDICTED 1/2/4/8-GPU series of the strong-scaling mode (bench.py --collection 10000), measured on ONE uncontended MI355X:
the phases that shard linearly (seeding, marker sets, seed tables, chaining) are timed once on the whole collection; the screen, which does not, is timed
per world size W through the public key-range entry points (skh_screen_part for part 0 of W + skh_screen_from_cells over the cells of all parts = what
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
    best = None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.screen_part(ss, 0, W)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        a, b = ctx.screen_from_cells(ss, allc, 0.0, True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        cur = ((t1 - t0) * 1e3, (t2 - t1) * 1e3)
        best = cur if best is None or sum(cur) < sum(best) else best
    assert len(a) == nch, (len(a), nch)
    scr[str(W)] = {"part_ms": best[0], "from_cells_ms": best[1], "cells_of_part_0": int(len(cells[0])), "cells_total": int(len(allc))}
out["screen_by_key_range_ms"] = scr
ctx.timings()
a, b = ctx.screen(ss, None, 0.0, 0, True)
out["screen_full_ms"] = ctx.timings()["screen_ms"]
print(json.dumps(out))
