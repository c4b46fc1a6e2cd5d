"""Per-rank cost of the distributed triangle's screen (rows = this rank's 1000 genomes, columns = all ranks' genomes) at simulated world sizes."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import skani_amd as sk
import bench
dev = torch.device("cuda:0")
ctx = sk.Context(0)
bases, contig_off, contig_genome, ng, _ = bench.make_genomes(torch, dev, np.arange(1000), mean_len=5_000_000, members=20)
gs = ctx.pack_buffer(None, contig_off, contig_genome, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
params = sk.SketchParams(125, 15, 1000, sk.SEED_AVX2)
ss = ctx.sketch_genomes(gs, params, genome_rank=np.arange(ng, dtype=np.uint32))
meta = ss.export_meta(); P, M, NC = ss.totals()
mk = np.zeros(M, np.uint64); ss.export_arrays(markers=mk)
for W in (1, 2, 4, 8):
    n = ng * W
    mo = np.concatenate([[0], np.cumsum(np.tile(np.diff(meta["marker_off"]), W))]).astype(np.uint64)
    co = np.concatenate([[0], np.cumsum(np.tile(np.diff(meta["contig_off"]), W))]).astype(np.uint64)
    gm = dict(pos_off=np.zeros(n + 1, np.uint64), marker_off=mo, contig_off=co, contig_lengths=np.tile(meta["contig_lengths"], W), total_len=np.tile(meta["total_len"], W),
              genome_rank=np.arange(n, dtype=np.uint32))
    # distinct markers per replica so that replicas do not match each other: xor a replica tag into the high bits
    allmk = np.concatenate([mk ^ (np.uint64(r) << np.uint64(36)) if r else mk for r in range(W)])
    allset = ctx.import_flat(params, gm, markers=allmk)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, r = ctx.screen_rows(allset, 0, ng, 0.0, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("world", W, "screen rows x cols", ng, n, "pairs", len(q), "ms", round(dt * 1e3, 2), flush=True)
    allset.close()
