"""Experiment: does seeding (VALU-bound) of one half of the genomes overlap with the table build (latency-bound) of the other half?
Two contexts (own streams) on one GPU, two host threads, each sketching 500 of the 1000 genomes, the second started with a delay; compared with one context sketching all 1000."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
import bench, skani_amd as sk
dev = torch.device("cuda", 0)
params = sk.SketchParams(125, 15, 1000, sk.SEED_AVX2)
c0, c1 = sk.Context(0), sk.Context(0)
sets = []
for a in (0, 500):
    bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, np.arange(a, a + 500))
    torch.cuda.synchronize()
    sets.append((bases, coff, cgen, ng))
g0 = c0.pack_buffer(None, sets[0][1], sets[0][2], 500, sk.SEED_AVX2, device_ptr=sets[0][0].data_ptr())
g1 = c1.pack_buffer(None, sets[1][1], sets[1][2], 500, sk.SEED_AVX2, device_ptr=sets[1][0].data_ptr())
allb, allo, allg, alln, _ = bench.make_genomes(torch, dev, np.arange(1000)); torch.cuda.synchronize()
gall = c0.pack_buffer(None, allo, allg, 1000, sk.SEED_AVX2, device_ptr=allb.data_ptr())
def one(ctx, gs, out, delay=0.0):
    if delay: time.sleep(delay)
    out.append(ctx.sketch_genomes(gs, params))
for rep in range(3):
    t = time.perf_counter(); s = c0.sketch_genomes(gall, params); t_all = time.perf_counter() - t; s.close()
    t = time.perf_counter(); a = c0.sketch_genomes(g0, params); b = c1.sketch_genomes(g1, params); t_seq = time.perf_counter() - t; a.close(); b.close()
    res = {}
    for delay in (0.0, 0.001, 0.002):
        o0, o1 = [], []
        th0 = threading.Thread(target=one, args=(c0, g0, o0)); th1 = threading.Thread(target=one, args=(c1, g1, o1, delay))
        t = time.perf_counter(); th0.start(); th1.start(); th0.join(); th1.join(); res[delay] = time.perf_counter() - t
        o0[0].close(); o1[0].close()
    print("all 1000 in one call %.2f ms | two halves one after the other %.2f ms | two threads, second delayed by 0 / 1 / 2 ms: %s" %
          (t_all * 1e3, t_seq * 1e3, " / ".join("%.2f" % (res[d] * 1e3) for d in sorted(res))))
