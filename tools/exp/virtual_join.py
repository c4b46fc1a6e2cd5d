#!/usr/bin/env python3
"""Experiment for the "one probe per position per cluster" join: what does the count pass cost when an enumerated sketch probes ONE table that
holds all 20 members of its clade instead of 9.5 tables of single partners?  The clade tables are made with what exists: the members' sketches exported,
concatenated (contigs of the members become contigs of one virtual genome) and imported as one genome per clade; every genome of the collection is then
chained against its clade's virtual genome with the ordinary kernels.  The anchors mean nothing (all members at once, band rule on the union) -- the point is
the memory behaviour of the count pass: one home-slot gather per position, every probe a hit, list heads of ~11 entries.  Run under rocprofv3 --kernel-trace;
compare join_count_kernel's time with the 2.15 ms of the per-pair join over the same 1000 genomes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import skani_amd as sk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
ctx = sk.Context(0)
bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, np.arange(n))
torch.cuda.synchronize()
gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
del bases
params = sk.SketchParams(125, 15, 1000, sk.SEED_AVX2)
ss = ctx.sketch_genomes(gs, params, genome_rank=np.arange(n, dtype=np.uint32))
recs = []
for cl in range(n // 20):
    seed, pos, cc, mk, cl_len = [], [], [], [], []
    nctg = 0
    for m in range(20):
        e = ss.export(cl * 20 + m)
        seed.append(e["seed"]); pos.append(e["pos"]); cc.append(e["ctgcanon"] + np.uint32(2 * nctg)); cl_len.append(e["contig_lengths"]); nctg += len(e["contig_lengths"])
    recs.append(dict(seed=np.concatenate(seed), pos=np.concatenate(pos), ctgcanon=np.concatenate(cc), markers=np.arange(3, dtype=np.uint64),
                     contig_lengths=np.concatenate(cl_len), total_len=int(np.concatenate(cl_len).sum())))
V = ctx.import_sketches(params, recs, genome_rank=np.arange(10_000_000, 10_000_000 + len(recs), dtype=np.uint32))
mp = sk.MapParams(learned_ani=True, compute_ci=True)
pr = np.arange(n, dtype=np.uint32) // 20; pq = np.arange(n, dtype=np.uint32)
for rep in range(3):
    ctx.timings(); t0 = time.perf_counter()
    res = ctx.chain_pairs(V, ss, pr, pq, mp)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("virtual clade tables: %d pairs (genome x its clade's 20-member table) chained in %.2f ms" % (n, dt * 1e3), ctx.timings()["chain_ms"], flush=True)
# the per-pair join of the same collection, for the same trace
i, j, r, nch = ctx.triangle(ss, mp)
print("per-pair triangle: %d pairs chained" % nch)
