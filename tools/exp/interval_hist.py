"""Histogram of candidate chain intervals / chunks per chained pair in the bench workload (sizes the greedy / finalize capacity classes)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
import skani_amd as sk
import bench
dev = torch.device("cuda:0"); ctx = sk.Context(0)
bases, contig_off, contig_genome, ng, _ = bench.make_genomes(torch, dev, np.arange(200), mean_len=5_000_000, members=20)
gs = ctx.pack_buffer(None, contig_off, contig_genome, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
ss = ctx.sketch_genomes(gs, sk.SketchParams())
a, b = ctx.screen(ss, None, 0.0, 0, True)
res, st = ctx.chain_pairs(ss, None, a, b, sk.MapParams(learned_ani=True), stats=True)
for f in ("n_intervals", "n_accepted", "n_chunks", "n_estimates", "n_anchors"):
    v = st[f].astype(np.int64)
    print(f, "min %d p10 %d p50 %d p90 %d max %d mean %.0f" % (v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max(), v.mean()))
v = st["n_intervals"]
print("intervals <=128 %d <=256 %d <=384 %d <=512 %d <=1024 %d >1024 %d of %d" % ((v <= 128).sum(), (v <= 256).sum(), (v <= 384).sum(), (v <= 512).sum(), (v <= 1024).sum(), (v > 1024).sum(), len(v)))
