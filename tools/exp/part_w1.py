#!/usr/bin/env python3
"""Where does a key-range part over ALL keys of 10,000 genomes spend its time, clade-ordered against shuffled?  (SKH_TRACE=1 marks of screen_partial_cells_dev.)
usage: SKH_TRACE=1 python tools/exp/part_w1.py [clade|shuffled] [free|keep|sleep]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import skani_amd as sk
order = sys.argv[1] if len(sys.argv) > 1 else "clade"
n = 10000
dev = torch.device("cuda", 0)
ctx = sk.Context(0)
canon = bench.genome_order(n, "shuffled")
ids = np.sort(canon) if order == "clade" else canon
bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, ids)
torch.cuda.synchronize()
gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
mode = sys.argv[2] if len(sys.argv) > 2 else "free"     # free: the 49 GB of ASCII genomes go back to the driver here (what bench.py does); keep: they stay allocated; sleep: freed, then 2 s of nothing
if mode != "keep":
    del bases; torch.cuda.empty_cache()
if mode == "sleep":
    time.sleep(2.0)
ss = ctx.sketch_genomes(gs, sk.SketchParams(125, 15, 1000, sk.SEED_AVX2), genome_rank=np.arange(n, dtype=np.uint32))
ctx.triangle(ss, sk.MapParams(learned_ani=True))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = ctx.screen_part(ss, 0, 1)
    torch.cuda.synchronize(); print(order, "rep", rep, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), len(c), file=sys.stderr)
