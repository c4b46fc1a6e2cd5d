"""Does chaining two halves of the pair list at the same time beat chaining them in one batch?  The headline collection's 9,500 pairs through
skh_chain_pairs: one context over all pairs, against two contexts (two host threads, own queues and arenas, the sketch set shared) over a half each."""
import sys, time, threading
import numpy as np
sys.path.insert(0, '.')
import torch
import bench as B
import skani_amd as sk
device = torch.device("cuda:0")
n = 1000
bases, coff, cgen, ng, _ = B.make_genomes(torch, device, np.arange(n))
torch.cuda.synchronize()
ctx = sk.Context(0); ctx2 = sk.Context(0)
params = sk.SketchParams(125, 15, 1000, sk.SEED_AVX2)
gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
ss = ctx.sketch_genomes(gs, params, genome_rank=np.arange(n, dtype=np.uint32))
pq, pr = ctx.screen(ss, None, 0.8)
pq = np.ascontiguousarray(pq, np.uint32); pr = np.ascontiguousarray(pr, np.uint32)
mp = sk.MapParams(learned_ani=True, compute_ci=True)
print("pairs", len(pq))
def timed(fn, reps=8):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
one = timed(lambda: ctx.chain_pairs(ss, None, pr, pq, mp))
h = len(pq) // 2
def both():
    out = [None, None]
    def run(c, lo, hi, k): out[k] = c.chain_pairs(ss, None, pr[lo:hi], pq[lo:hi], mp)
    t = threading.Thread(target=run, args=(ctx2, h, len(pq), 1)); t.start(); run(ctx, 0, h, 0); t.join()
    return out
two = timed(both)
seq = timed(lambda: (ctx.chain_pairs(ss, None, pr[:h], pq[:h], mp), ctx.chain_pairs(ss, None, pr[h:], pq[h:], mp)))
print("one batch %.3f ms; two halves one after the other %.3f ms; two halves at the same time (two contexts, two threads) %.3f ms" % (one, seq, two))
a = ctx.chain_pairs(ss, None, pr, pq, mp); b = both()
assert a.tobytes() == b[0].tobytes() + b[1].tobytes()
# more parts, and a second half that starts late (its count pass beside the first half's DP)
ctxs = [ctx, ctx2, sk.Context(0), sk.Context(0)]
for parts in (3, 4):
    cuts = [len(pq) * k // parts for k in range(parts + 1)]
    def many():
        th = [threading.Thread(target=lambda k=k: ctxs[k].chain_pairs(ss, None, pr[cuts[k]:cuts[k + 1]], pq[cuts[k]:cuts[k + 1]], mp)) for k in range(1, parts)]
        for t in th: t.start()
        ctxs[0].chain_pairs(ss, None, pr[cuts[0]:cuts[1]], pq[cuts[0]:cuts[1]], mp)
        for t in th: t.join()
    print("%d parts at the same time: %.3f ms" % (parts, timed(many)))
for delay_ms in (0.5, 1.0, 1.5, 2.0):
    def late():
        def run(): time.sleep(0); t0 = time.perf_counter()
        def second():
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < delay_ms: pass
            ctx2.chain_pairs(ss, None, pr[h:], pq[h:], mp)
        t = threading.Thread(target=second); t.start(); ctx.chain_pairs(ss, None, pr[:h], pq[:h], mp); t.join()
    print("second half %.1f ms late: %.3f ms" % (delay_ms, timed(late)))
