"""pack_kernel variants (tools/exp/build_variants.py pack_seed.hip PACK_ROUNDS_N 4 8 16) on device-resident ASCII: ms per 4.9 Gbases from the context's pack timer."""
import sys, glob, os
import numpy as np
sys.path.insert(0, '.')
import torch
import skani_amd as sk
from skani_amd import _binding as B
n_genomes, L = 1000, 4_900_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
codes = torch.randint(0, 4, (n_genomes * L,), dtype=torch.uint8, device="cuda", generator=g)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
bases = torch.empty_like(codes)
for at in range(0, codes.numel(), 1 << 28): bases[at:at + (1 << 28)] = lut[codes[at:at + (1 << 28)].long()]
del codes
contig_off = np.arange(n_genomes + 1, dtype=np.uint64) * L
contig_genome = np.arange(n_genomes, dtype=np.uint32)
for path in sorted(glob.glob("tools/exp/variants/libskani_hip_*.so")):
    lib = B.load(path)
    ctx = sk.Context(0, lib=lib)
    res = []
    for it in range(4):
        t0 = ctx.timings()["pack_ms"]
        gs = ctx.pack_buffer(None, contig_off, contig_genome, n_genomes, sk.SEED_AVX2, device_ptr=bases.data_ptr())
        torch.cuda.synchronize()
        res.append(ctx.timings()["pack_ms"] - t0); gs.close()
    print(os.path.basename(path), ["%.3f" % x for x in res], flush=True)
    ctx.close()
