"""PCIe-inclusive ingest: host ASCII bases -> skh_sketch_batch (H2D + pack + seed + tables).  Reports GB/s of ASCII handed over."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import skani_amd as sk

ctx = sk.Context(0)
n_genomes, L = 200, 5_000_000
rng = np.random.default_rng(1)
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_genomes * L, dtype=np.uint8)]
contig_off = np.arange(n_genomes + 1, dtype=np.uint64) * L
contig_genome = np.arange(n_genomes, dtype=np.uint32)
for it in range(3):
    t0 = time.perf_counter()
    gs = ctx.pack_buffer(bases, contig_off, contig_genome, n_genomes, sk.SEED_AVX2)
    t1 = time.perf_counter()
    ss = ctx.sketch_genomes(gs, sk.SketchParams())
    t2 = time.perf_counter()
    print("pack (H2D + 2-bit pack) %.1f ms = %.1f GB/s of ASCII; sketch %.1f ms; total %.1f GB/s" %
          ((t1 - t0) * 1e3, bases.nbytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bases.nbytes / (t2 - t0) / 1e9), flush=True)
    gs.close(); ss.close()
