#!/usr/bin/env python3
"""bench.py -- skani triangle hot path (sketch -> screen -> chain -> ANI/AF) on MI355X.

Metric (BASELINE.json): genome-pairs/sec for an all-vs-all triangle of synthetic ~5 Mbp bacterial genomes.
One "step" = one full pass of the hot path over the resident batch: FracMinHash seeding of every genome from
2-bit packed bases already in HBM, sketch-table construction, marker screen of all N(N-1)/2 pairs, chaining +
ANI/AF (+ learned ANI) of every pair that passes the screen.  Default -c 125 -k 15 -m 1000 -s 80.

Workload at N GPUs (weak scaling): 1000 genomes per GPU in clades of 20 (config 4's
10k-genome triangle is the 8-GPU point at 8000 genomes).  Every rank sketches its own 1000 genomes, the sketches
are all-gathered (RCCL), each rank screens the full set and chains its round-robin share of the passing pairs,
results are gathered on rank 0.  value = N_total*(N_total-1)/2 / step time.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (seeding kernel, HBM bound per
SURVEY 8d: 0.354 algorithmic bytes/base) and `cpu_baseline` (the C++ oracle = a port of the reference algorithms,
timed on this box's host cores on a bounded sample; the Rust reference cannot be built here).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C, K, M = 125, 15, 1000
CLADE = 20
SEED0 = 0x5EED0000


def make_genomes(torch, device, first_clade, n_clades, members=CLADE, mean_len=5_000_000, keep_ascii_clades=0):
    """Deterministic synthetic genomes generated ON THE GPU: clade root = i.i.d. ACGT of length
    U(0.9,1.1)*mean_len; member = root with substitution rate U(0.005,0.08), 0-5 deletions of 10-50 kb, split into
    1-20 contigs (>= 10 kb).  Returns (ascii uint8 device tensor, contig_off, contig_genome, n_genomes, host copies)."""
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    pieces, contig_off, contig_genome = [], [0], []
    host_genomes = []
    g_idx = 0
    for cl in range(first_clade, first_clade + n_clades):
        gen = torch.Generator(device=device); gen.manual_seed(SEED0 + cl)
        cpu_rng = np.random.default_rng(SEED0 + cl)
        L = int(cpu_rng.integers(int(mean_len * 0.9), int(mean_len * 1.1) + 1))
        root = torch.randint(0, 4, (L,), dtype=torch.uint8, device=device, generator=gen)
        for m in range(members):
            d = float(cpu_rng.uniform(0.005, 0.08))
            mask = torch.rand(L, device=device, generator=gen) < d
            sub = torch.randint(1, 4, (L,), dtype=torch.uint8, device=device, generator=gen)
            codes = torch.where(mask, (root + sub) & 3, root)
            keep = torch.ones(L, dtype=torch.bool, device=device)
            for _ in range(int(cpu_rng.integers(0, 6))):
                dl = int(cpu_rng.integers(10_000, 50_001)) * mean_len // 5_000_000 if mean_len < 5_000_000 else int(cpu_rng.integers(10_000, 50_001))
                dl = max(dl, 1)
                s = int(cpu_rng.integers(0, max(1, L - dl)))
                keep[s:s + dl] = False
            codes = codes[keep]
            asc = lut[codes.long()]
            n = asc.numel()
            min_ctg = 10_000 if mean_len >= 1_000_000 else 2_000
            n_ctg = int(cpu_rng.integers(1, 21))
            n_ctg = max(1, min(n_ctg, n // (2 * min_ctg)))
            if n_ctg > 1:
                cuts = np.sort(cpu_rng.choice(np.arange(1, n // min_ctg), n_ctg - 1, replace=False)) * min_ctg
            else:
                cuts = np.array([], dtype=np.int64)
            bounds = [0] + [int(c) for c in cuts] + [n]
            for a, b in zip(bounds[:-1], bounds[1:]):
                contig_off.append(contig_off[-1] + (b - a)); contig_genome.append(g_idx)
            pieces.append(asc)
            if cl - first_clade < keep_ascii_clades:
                h = asc.cpu().numpy().tobytes()
                host_genomes.append([("c%d" % i, h[a:b]) for i, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))])
            g_idx += 1
    bases = torch.cat(pieces)
    return bases, np.array(contig_off, np.uint64), np.array(contig_genome, np.uint32), g_idx, host_genomes


def make_queries(torch, device, clades, mean_len=5_000_000):
    """One fresh member (2% substitutions) of each listed clade: the clade root is regenerated from its seed."""
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    pieces, contig_off, contig_genome = [], [0], []
    for qi, cl in enumerate(clades):
        gen = torch.Generator(device=device); gen.manual_seed(SEED0 + int(cl))
        cpu_rng = np.random.default_rng(SEED0 + int(cl))
        L = int(cpu_rng.integers(int(mean_len * 0.9), int(mean_len * 1.1) + 1))
        root = torch.randint(0, 4, (L,), dtype=torch.uint8, device=device, generator=gen)
        g2 = torch.Generator(device=device); g2.manual_seed(SEED0 + 10_000_000 + qi)
        mask = torch.rand(L, device=device, generator=g2) < 0.02
        sub = torch.randint(1, 4, (L,), dtype=torch.uint8, device=device, generator=g2)
        pieces.append(lut[torch.where(mask, (root + sub) & 3, root).long()])
        contig_off.append(contig_off[-1] + L); contig_genome.append(qi)
    return torch.cat(pieces), np.array(contig_off, np.uint64), np.array(contig_genome, np.uint32), len(clades)


def run_search(args, torch, sk, ctx, device):
    """BASELINE config 5 (optional workload): queries vs a pre-sketched database RESIDENT in HBM, --medium preset (c=70)."""
    params = sk.SketchParams(args.c, K, M, sk.SEED_AVX2)
    shard = 1000
    n_db = args.db_genomes
    shards = []
    t0 = time.perf_counter()
    for a in range(0, n_db, shard):
        n = min(shard, n_db - a)
        bases, coff, cgen, ng, _ = make_genomes(torch, device, a // CLADE, (n + CLADE - 1) // CLADE, members=CLADE, mean_len=args.mean_len)
        torch.cuda.synchronize()
        gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
        del bases
        shards.append(ctx.sketch_genomes(gs, params, genome_rank=np.arange(a, a + ng, dtype=np.uint32)))
        gs.close(); torch.cuda.empty_cache()
    db = sk.SketchDB(shards)
    build_s = time.perf_counter() - t0
    rng = np.random.default_rng(12345)
    qclades = rng.integers(0, max(1, n_db // CLADE), args.queries)
    qb, qoff, qgen, nq = make_queries(torch, device, qclades, args.mean_len)
    torch.cuda.synchronize()
    gq = ctx.pack_buffer(None, qoff, qgen, nq, sk.SEED_AVX2, device_ptr=qb.data_ptr())
    del qb
    qs = ctx.sketch_genomes(gq, params, genome_rank=np.arange(10_000_000, 10_000_000 + nq, dtype=np.uint32))
    gq.close()
    mem_gb = torch.cuda.mem_get_info(device)
    for _ in range(args.warmup):
        sk.search(ctx, db, qs, n_query_files=nq)
    ctx.timings()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        q, r, res = sk.search(ctx, db, qs, n_query_files=nq)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    tm = ctx.timings()
    own = (r // CLADE) == qclades[q]
    print(json.dumps({"metric": "search queries/sec vs resident sketch DB", "value": nq / dt, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                      "config": {"workload": "skani search: %d synthetic queries vs %d-genome DB (c=%d) resident in HBM" % (nq, n_db, args.c), "db_genomes": n_db,
                                 "db_shards": len(shards), "queries": nq, "hits": int(len(q)), "hits_in_own_clade": int(own.sum()), "db_build_s": build_s,
                                 "hbm_used_gb": (mem_gb[1] - mem_gb[0]) / 1e9},
                      "phase_ms_per_step": {k: tm[k] / args.steps for k in ("screen_ms", "chain_ms")}, "roofline": None, "cpu_baseline": None}))


def cpu_baseline(host_genomes, n_full, chained_full, threads, gpu_result=None):
    """Times the oracle (port of the reference algorithms; kind = 'port') on the sample with all host cores and, when the
    GPU triangle result is given, reports the metric's "ANI delta vs ref" on the sample's pairs."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py as ora
    names = ["s%04d.fa" % i for i in range(len(host_genomes))]
    # regression.rs:8-28: learned ANI only for c >= 70, table chosen by |c-125| < |c-200|
    model = None
    if C >= 70:
        model = ora.Model(os.path.join(ROOT, "skani_amd", "data", "gbdt_c125.bin" if abs(C - 125) < abs(C - 200) else "gbdt_c200.bin"))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:     # ctypes releases the GIL: genomes are sketched in parallel like file_io.rs:147
        sks = list(ex.map(lambda a: ora.sketch_records(a[1], C, K, M, names[a[0]], 1), enumerate(host_genomes)))
    t1 = time.perf_counter()
    oi, oj, res, n_chained, n_pass = ora.triangle(sks, model=model, threads=threads)
    t2 = time.perf_counter()
    n = len(host_genomes); pairs = n * (n - 1) // 2
    bases = sum(len(s) for g in host_genomes for _, s in g)
    sketch_s, chain_s = t1 - t0, t2 - t1
    # per-unit costs -> the same workload the GPU ran (sketching is linear in genomes, chaining in chained pairs)
    est_full = sketch_s / n * n_full + chain_s / max(n_chained, 1) * chained_full
    delta = None
    if gpu_result is not None:
        gi, gj, gres = gpu_result
        sel = (gi < n) & (gj < n)
        gkeys = gi[sel].astype(np.int64) * n + gj[sel]; okeys = oi.astype(np.int64) * n + oj
        same = len(gkeys) == len(okeys) and bool(np.array_equal(gkeys, okeys))
        delta = {"pairs_compared": int(len(okeys)), "same_pair_set": same}
        if same and len(okeys):
            g = gres[sel]
            for f in ("ani", "af_ref", "af_query"):
                delta["max_abs_d_" + f] = float(np.max(np.abs(g[f].astype(np.float64) - res[f].astype(np.float64))))
            delta["int_fields_equal"] = bool(all(np.array_equal(g[f], res[f]) for f in ("avg_chain_int_len", "total_bases_covered", "num_contigs_q", "num_contigs_r")))
    return {"value": (n_full * (n_full - 1) // 2) / est_full, "unit": "genome-pairs/s", "cores": threads, "kind": "port", "delta_vs_oracle": delta,
            "sample": "%d synthetic genomes (%d clades of %d, %.0f Mbp): oracle sketch %.2f s + screen/chain of %d pairs (%d chained) %.2f s on %d threads; "
                      "value = full-workload pairs / (per-genome sketch cost x %d + per-chained-pair cost x %d)" %
                      (n, n // CLADE, CLADE, bases / 1e6, sketch_s, pairs, n_chained, chain_s, threads, n_full, chained_full),
            "sample_pairs_per_s": pairs / (t2 - t0), "sample_sketch_mbases_per_s": bases / 1e6 / sketch_s,
            "sample_chained_pairs_per_s": n_chained / chain_s if chain_s > 0 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes-per-gpu", type=int, default=1000)
    ap.add_argument("--mean-len", type=int, default=5_000_000)
    ap.add_argument("--cpu-clades", type=int, default=6, help="clades (x20 genomes) in the CPU-baseline sample; 0 disables")
    ap.add_argument("--no-ci", action="store_true")
    ap.add_argument("--c", type=int, default=125, help="-c compression factor (presets: 30 slow, 70 medium, 125 default, 200 fast)")
    ap.add_argument("--clade", type=int, default=20, help="genomes per clade (= genomes-per-gpu gives the dense single-clade variant)")
    ap.add_argument("--workload", default="triangle", choices=["triangle", "search"], help="triangle = the headline metric; search = BASELINE config 5 (optional)")
    ap.add_argument("--db-genomes", type=int, default=10000)
    ap.add_argument("--queries", type=int, default=200)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import skani_amd as sk
    global C, CLADE
    C, CLADE = args.c, args.clade

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    ctx = sk.Context(local)
    if args.workload == "search":
        if world > 1:
            raise SystemExit("the search workload is single-GPU")
        if args.c == 125:
            args.c = C = 70            # --medium, BASELINE config 5
        run_search(args, torch, sk, ctx, device)
        return

    n_local = args.genomes_per_gpu
    assert n_local % CLADE == 0
    clades_local = n_local // CLADE
    n_total = n_local * world
    keep = args.cpu_clades if rank == 0 and world == 1 else 0
    bases, contig_off, contig_genome, ng, host_genomes = make_genomes(torch, device, rank * clades_local, clades_local, mean_len=args.mean_len,
                                                                      members=CLADE, keep_ascii_clades=min(keep, clades_local))
    torch.cuda.synchronize()
    gs = ctx.pack_buffer(None, contig_off, contig_genome, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
    total_bases_local = int(contig_off[-1])
    del bases
    torch.cuda.empty_cache()
    params = sk.SketchParams(C, K, M, sk.SEED_AVX2)
    mp = sk.MapParams(learned_ani=sk.use_learned_ani(C), compute_ci=not args.no_ci)
    ctx.timings()

    from skani_amd.distributed import distributed_triangle
    last = {}

    def step():
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=np.arange(rank * n_local, (rank + 1) * n_local, dtype=np.uint32))
        i, j, res, n_chained = distributed_triangle(ctx, ss_local, params, mp, dist, rank, world, torch=torch, device=device)
        last["result"] = (i, j, res)
        return (len(i) if i is not None else 0), n_chained

    for _ in range(args.warmup):
        step()
    ctx.timings()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kept = chained = 0
    for _ in range(args.steps):
        kept, chained = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        c2 = torch.tensor([chained if rank != 0 else 0, kept], dtype=torch.int64, device=device)
        if rank == 0:
            c2[0] = chained      # rank 0 already holds the total (gathered); the others report their own share
        dist.broadcast(c2, src=0); chained, kept = int(c2[0]), int(c2[1])
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms_per_step = dt / args.steps * 1e3
    pairs = n_total * (n_total - 1) // 2
    value = pairs / (dt / args.steps)
    # roofline of the seeding kernel: algorithmic bytes = 0.25 B/base packed read + 12/c B seeds + 8/m B markers
    alg_bytes_per_base = 0.25 + 12.0 / C + 8.0 / M
    launches = max(tm["seed_kernel_launches"], 1)
    seed_ms_per_launch = tm["seed_kernel_ms"] / launches
    bytes_per_launch = alg_bytes_per_base * total_bases_local * args.steps / launches
    achieved = bytes_per_launch / (seed_ms_per_launch * 1e-3) / 1e9 if seed_ms_per_launch > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "seed_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "genome-pairs/sec (triangle, ~5 Mbp genomes)", "value": value, "unit": "genome-pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "skani triangle over %d synthetic ~%.1f Mbp genomes (clades of %d, 0.5-8%% divergence), -c %d -k %d -m %d -s 80, learned ANI on"
                               % (n_total, args.mean_len / 1e6, CLADE, C, K, M),
                   "genomes": n_total, "genomes_per_gpu": n_local, "bases_per_gpu": total_bases_local, "pairs": pairs, "chained_pairs": chained,
                   "kept_pairs": kept, "parallelism": "genomes block-sharded; every rank screens and chains the pairs of its own rows; markers all-gathered"},
        "phase_ms_per_step": {k: tm[k] / args.steps for k in ("seed_ms", "sketch_build_ms", "screen_ms", "chain_ms")},
        "roofline": {"kernel": "seed_tiles_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": traffic, "bytes_per_launch": bytes_per_launch, "ms_per_launch": seed_ms_per_launch, "launches_per_step": launches / args.steps,
                     "note": "0.354 algorithmic B/base; kernel is bound by VALU issue (23 instructions per window, 64-bit hash mix per base), see DESIGN.md and profiles/r01_valu_rates.md"},
    }
    # the chaining pipeline against the north star's algorithmic figure: both sketches of a chained pair read once, 12 B per position
    # (~0.96 MB per pair of 5 Mbp genomes at c=125)
    chain_s = tm["chain_ms"] / args.steps * 1e-3
    if chain_s > 0 and chained:
        chain_bytes = 12.0 * 2.0 * (total_bases_local / max(n_local, 1) / C) * (chained / world)
        out["roofline_chain"] = {"stage": "join + chunk + chain + select + estimate (8 kernels)", "bound": "hbm", "achieved": chain_bytes / chain_s / 1e9, "peak": 8000.0,
                                 "unit": "GB/s", "frac": chain_bytes / chain_s / 1e9 / 8000.0, "bytes_per_step": chain_bytes,
                                 "note": "irregular, latency-bound stages: per-kernel traffic and occupancy in profiles/r01_pmc_v9.md"}
    if host_genomes:
        out["cpu_baseline"] = cpu_baseline(host_genomes, n_total, chained, os.cpu_count() or 1, last.get("result"))
        # skani's default thread count (-t 3, cli.rs:243) on one clade of the same sample, for a like-for-like default
        few = cpu_baseline(host_genomes[:CLADE], n_total, chained, 3, None)
        out["cpu_baseline"]["default_threads"] = {"cores": 3, "value": few["value"], "sample": few["sample"]}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
