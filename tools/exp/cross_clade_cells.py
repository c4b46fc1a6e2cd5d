"""How many cells of the triangle's count matrix lie OUTSIDE the clades of the synthetic collection (pairs of unrelated genomes that share a marker), and with what counts?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench, skani_amd as sk
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, np.arange(n)); torch.cuda.synchronize()
ctx = sk.Context(0)
gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
ss = ctx.sketch_genomes(gs, sk.SketchParams(125, 15, 1000, sk.SEED_AVX2), genome_rank=np.arange(n, dtype=np.uint32))
cells = ctx.screen_part(ss, 0, 1)
i, j, c = ctx.unpack_cells(cells)
cross = (i // 20) != (j // 20)
print("cells", len(c), "inside clades", int((~cross).sum()), "across clades", int(cross.sum()), "counts across:", np.bincount(c[cross])[:8].tolist() if cross.any() else [])
for a, b, k in list(zip(i[cross], j[cross], c[cross]))[:10]:
    ma, mb = ss.export(int(a))["markers"], ss.export(int(b))["markers"]
    sh = np.intersect1d(ma, mb)
    def kmer(v): return "".join("ACGT"[(int(v) >> (2 * (20 - x))) & 3] for x in range(21))
    print(int(a), int(b), int(k), [kmer(v) for v in sh[:3]])
