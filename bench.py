#!/usr/bin/env python3
"""bench.py -- skani triangle hot path (sketch -> screen -> chain -> ANI/AF) on MI355X.

Metric (BASELINE.json): genome-pairs/sec for an all-vs-all triangle of synthetic ~5 Mbp bacterial genomes.
One "step" = one full pass of the hot path over the resident batch: FracMinHash seeding of every genome from
2-bit packed bases already in HBM, sketch-table construction, marker screen of all N(N-1)/2 pairs, chaining +
ANI/AF (+ learned ANI) of every pair that passes the screen.  Default -c 125 -k 15 -m 1000 -s 80.

Workload: N = 1: 1000 genomes in clades of 20 (BASELINE config 3).  N > 1 (weak scaling, the default): 1250 genomes per GPU -- 10,000 on
8 GPUs is BASELINE config 4 -- in SHUFFLED order (files sort by name, not by clade: file_io.rs:250), so the members of a clade sit on different
GPUs.  `--collection 10000` is the STRONG-scaling mode: config 4's 10,000 shuffled genomes at every N (10000 / N per rank; N = 1 fits one GPU).
The pair count grows with the square of the collection while the work (bases seeded, pairs chained) grows linearly, so the line also carries
`bases_per_s_per_gpu` and `chained_pairs_per_s_per_gpu`: the rates that compare across N and across the two modes.
`--one-device` puts all ranks on cuda:0 (host collectives over gloo): the whole multi-rank branch of this file on a one-GPU box.  Every rank sketches its own genomes; skh_triangle_distributed (csrc/dist.hip, RCCL below the C ABI) all-gathers the marker sets,
screens a share of the rows on every rank, assigns the candidate pairs to ranks cluster by cluster (balanced, order-independent), moves
exactly the sketches that are needed point-to-point and gathers the results.  value = N_total (N_total - 1) / 2 / step time.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (seeding kernel, HBM bound per SURVEY 8d: 0.354 algorithmic
bytes/base; plus the VALU issue fraction that actually binds it) and `cpu_baseline` (the C++ oracle = a port of the reference algorithms,
run on this box's host cores on the same, full workload at several thread counts -- about 20 s in all; the best point is the value; the Rust
reference cannot be built here).
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
ROOT_CODES = None       # --root-fasta: a device tensor of 2-bit codes that every clade's root is taken from (a real genome) instead of i.i.d. bases


def genome_order(n_total, order):
    """Global genome index -> canonical id (clade * CLADE + member).  'clade': the collection is listed clade by clade; 'shuffled': in the
    order a directory of unrelated file names would sort (file_io.rs:250 sorts by NAME) -- members of a clade end up on different ranks."""
    if order == "clade":
        return np.arange(n_total, dtype=np.int64)
    return np.random.default_rng(SEED0 ^ 0x0F11E5).permutation(n_total).astype(np.int64)


def make_genomes(torch, device, wanted, members=CLADE, mean_len=5_000_000, keep_host=False):
    """Deterministic synthetic genomes generated ON THE GPU: clade root = i.i.d. ACGT of length U(0.9,1.1)*mean_len; member = root
    with substitution rate U(0.005,0.08), 0-5 deletions of 10-50 kb, split into 1-20 contigs (>= 10 kb).  `wanted` lists canonical ids
    (clade * members + member) in the order the caller wants them; a genome is a pure function of its id (the clade's random streams are
    replayed up to the member), so any rank can generate any subset.  Returns (ascii uint8 device tensor, contig_off, contig_genome,
    n_genomes, host copies [list of (name, uint8 array) per genome] when keep_host: True = of every genome, a boolean mask over `wanted` = of those)."""
    on_host = None if keep_host is False or keep_host is None else (np.ones(len(wanted), bool) if keep_host is True else np.asarray(keep_host, bool))
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    wanted = [int(x) for x in wanted]
    by_clade = {}
    for pos, cid in enumerate(wanted):
        by_clade.setdefault(cid // members, {})[cid % members] = pos
    pieces, bounds_of, host_of = [None] * len(wanted), [None] * len(wanted), [None] * len(wanted)
    min_ctg = 10_000 if mean_len >= 1_000_000 else 2_000
    for cl in sorted(by_clade):
        want = by_clade[cl]
        gen = torch.Generator(device=device); gen.manual_seed(SEED0 + cl)
        cpu_rng = np.random.default_rng(SEED0 + cl)
        L = int(cpu_rng.integers(int(mean_len * 0.9), int(mean_len * 1.1) + 1))
        if ROOT_CODES is not None:
            root = ROOT_CODES; L = int(root.numel())
        else:
            root = torch.randint(0, 4, (L,), dtype=torch.uint8, device=device, generator=gen)
        for m in range(max(want) + 1):
            d = float(cpu_rng.uniform(0.005, 0.08))
            mask = torch.rand(L, device=device, generator=gen) < d
            sub = torch.randint(1, 4, (L,), dtype=torch.uint8, device=device, generator=gen)
            dels = []
            for _ in range(int(cpu_rng.integers(0, 6))):
                dl = int(cpu_rng.integers(10_000, 50_001)) * mean_len // 5_000_000 if mean_len < 5_000_000 else int(cpu_rng.integers(10_000, 50_001))
                dl = max(dl, 1)
                st = int(cpu_rng.integers(0, max(1, L - dl)))
                dels.append((st, min(L, st + dl)))
            covered, end = 0, 0                                   # length of the union of the deleted intervals
            for a_, b_ in sorted(dels):
                if b_ > end:
                    covered += b_ - max(a_, end); end = b_
            n = L - covered
            n_ctg = int(cpu_rng.integers(1, 21))
            n_ctg = max(1, min(n_ctg, n // (2 * min_ctg)))
            if n_ctg > 1:
                cuts = np.sort(cpu_rng.choice(np.arange(1, n // min_ctg), n_ctg - 1, replace=False)) * min_ctg
            else:
                cuts = np.array([], dtype=np.int64)
            if m not in want:
                continue
            codes = torch.where(mask, (root + sub) & 3, root)
            if dels:
                keep = torch.ones(L, dtype=torch.bool, device=device)
                for a_, b_ in dels:
                    keep[a_:b_] = False
                codes = codes[keep]
            asc = lut[codes.long()]
            assert asc.numel() == n
            pos = want[m]
            pieces[pos] = asc
            bounds_of[pos] = [0] + [int(c) for c in cuts] + [n]
            if on_host is not None and on_host[pos]:
                h = asc.cpu().numpy()
                host_of[pos] = [("c%d" % i, h[a_:b_]) for i, (a_, b_) in enumerate(zip(bounds_of[pos][:-1], bounds_of[pos][1:]))]
    contig_off, contig_genome = [0], []
    for g, bnd in enumerate(bounds_of):
        for a_, b_ in zip(bnd[:-1], bnd[1:]):
            contig_off.append(contig_off[-1] + (b_ - a_)); contig_genome.append(g)
    bases = torch.cat(pieces) if pieces else torch.zeros(1, dtype=torch.uint8, device=device)
    return bases, np.array(contig_off, np.uint64), np.array(contig_genome, np.uint32), len(wanted), (host_of if on_host is not None else [])


def make_queries(torch, device, clades, mean_len=5_000_000, first=0):
    """One fresh member (2% substitutions) of each listed clade: the clade root is regenerated from its seed.  Query number first + x belongs to clades[x]."""
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    pieces, contig_off, contig_genome = [], [0], []
    for qi, cl in enumerate(clades, start=first):
        gen = torch.Generator(device=device); gen.manual_seed(SEED0 + int(cl))
        cpu_rng = np.random.default_rng(SEED0 + int(cl))
        L = int(cpu_rng.integers(int(mean_len * 0.9), int(mean_len * 1.1) + 1))
        root = torch.randint(0, 4, (L,), dtype=torch.uint8, device=device, generator=gen)
        g2 = torch.Generator(device=device); g2.manual_seed(SEED0 + 10_000_000 + qi)
        mask = torch.rand(L, device=device, generator=g2) < 0.02
        sub = torch.randint(1, 4, (L,), dtype=torch.uint8, device=device, generator=g2)
        pieces.append(lut[torch.where(mask, (root + sub) & 3, root).long()])
        contig_off.append(contig_off[-1] + L); contig_genome.append(qi - first)
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
        bases, coff, cgen, ng, _ = make_genomes(torch, device, np.arange(a, a + n), members=CLADE, mean_len=args.mean_len)
        torch.cuda.synchronize()
        gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
        del bases
        shards.append(ctx.sketch_genomes(gs, params, genome_rank=np.arange(a, a + ng, dtype=np.uint32), compact=not getattr(args, "no_compact", False)))   # SKH_SKETCH_COMPACT: a resident database
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
    torch.cuda.empty_cache()
    live_b, idle_b = ctx.device_memory(trim=True)                    # (the library's idle cache of freed build scratch handed back: the database itself is what stays)
    mem_gb = torch.cuda.mem_get_info(device)
    for _ in range(args.warmup):
        sk.search(ctx, db, qs, n_query_files=nq)
    ctx.timings()
    host_times = {} if os.environ.get("BENCH_STEP_TIMES") else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        q, r, res = sk.search(ctx, db, qs, n_query_files=nq, host_times=host_times)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    tm = ctx.timings()
    if host_times is not None:
        print("host view of a search step (ms): %s; library timers: %s" % ({k: round(1e3 * v / args.steps, 3) for k, v in host_times.items()},
                                                                           {k: round(v / args.steps, 3) for k, v in tm.items() if k.endswith("_ms") and v}), file=sys.stderr)
    own = (r // CLADE) == qclades[q]
    # ---- roofline: the chaining stage dominates the step; SURVEY 8d's figure: both sketches of a chained pair read once, 12 B per seed position (positions = bases / c);
    # the screen: 8 B per marker of the database and of the queries, read once
    pos_per_genome = args.mean_len / args.c
    chain_s = tm["chain_ms"] / args.steps * 1e-3; screen_s = tm["screen_ms"] / args.steps * 1e-3
    n_hits_chained = int(len(q))                                        # (every screened pair of this workload is kept: all hits lie in the query's clade)
    chain_bytes = 12.0 * 2.0 * pos_per_genome * n_hits_chained
    screen_bytes = 8.0 * (n_db + nq) * (args.mean_len / M)
    roof = {"stage": "join + chunk + chain + select + estimate over the screened (query, reference) pairs", "bound": "hbm", "achieved": chain_bytes / chain_s / 1e9 if chain_s > 0 else 0.0,
            "peak": 8000.0, "unit": "GB/s", "frac": chain_bytes / chain_s / 1e9 / 8000.0 if chain_s > 0 else 0.0, "traffic": None, "bytes_per_step": chain_bytes, "ms": chain_s * 1e3,
            "note": "12 B x (positions of query + reference) per chained pair (SURVEY 8d); irregular, latency-bound stages -- the same kernels as the triangle's chaining, whose counters are in profiles/r05_pmc.md",
            "screen": {"bound": "hbm", "achieved": screen_bytes / screen_s / 1e9 if screen_s > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                       "frac": screen_bytes / screen_s / 1e9 / 8000.0 if screen_s > 0 else 0.0, "bytes_per_step": screen_bytes, "ms": screen_s * 1e3,
                       "note": "8 B per marker of the database and the queries; the database's sorted incidence list is cached, a step sorts the queries' and probes"}}
    cpu = None
    if getattr(args, "cpu_queries", 0):
        try:
            cpu = search_cpu_baseline(args, torch, sk, device, db, qclades, q, r, res, n_db, nq)
        except Exception as e:                                           # the line must not depend on the oracle leg
            cpu = {"error": repr(e)}
    db.close(); qs.close()
    return ({"metric": "search queries/sec vs resident sketch DB", "value": nq / dt, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                      "config": {"workload": "skani search: %d synthetic queries vs %d-genome DB (c=%d) resident in HBM" % (nq, n_db, args.c), "db_genomes": n_db,
                                 "db_shards": len(shards), "queries": nq, "hits": int(len(q)), "hits_in_own_clade": int(own.sum()), "db_build_s": build_s,
                                 "hbm_used_gb": (mem_gb[1] - mem_gb[0]) / 1e9, "library_live_gb": live_b / 1e9,
                                 "bytes_per_seed_position": live_b / max(1.0, n_db * (args.mean_len / args.c)), "compact_shards": not getattr(args, "no_compact", False)},
                      "phase_ms_per_step": {k: tm[k] / args.steps for k in ("screen_ms", "chain_ms")}, "roofline": roof, "cpu_baseline": cpu},
            (q, r, res, qclades))


def search_cpu_baseline(args, torch, sk, device, db, qclades, q, r, res, n_db, nq):
    """The CPU side of the search metric on a BOUNDED sample: `--cpu-queries` of the queries against a sample of the database -- the queries' own clades (every
    reference the GPU chained for them) + other clades up to 100 -- through the oracle's search loop (ora_search: search.rs:97-200 with every reference resident;
    inverted marker index, screen_refs_indices, chain_seeds, ani > 0.5).  The reference sketches are the GPU database's own, exported (sketching a database is
    `skani sketch`'s job, not the search's); the queries are sketched by the oracle from their bytes (not timed: the GPU step takes sketched queries too).
    Scaled to the workload: index build x (database / sample refs), screen x (queries / sampled queries) (a probe's cost goes with the query's markers and the
    postings it meets, i.e. with its clade, not with the database), chain x (hits / sampled hits)."""
    from oracle import oracle_py as ora
    nsq = min(int(args.cpu_queries), nq)
    pick = np.unique(np.linspace(0, nq - 1, nsq).astype(np.int64))
    clades = list(dict.fromkeys(int(qclades[x]) for x in pick))
    n_cl = max(1, n_db // CLADE)
    for cl in (np.arange(100, dtype=np.int64) * n_cl // 100).tolist():             # other clades, evenly from the database
        if len(clades) >= 100: break
        if cl not in clades: clades.append(cl)
    ref_ids = np.sort(np.concatenate([np.arange(cl * CLADE, min(n_db, (cl + 1) * CLADE)) for cl in clades]))
    orefs = []
    for g in ref_ids:
        sh = int(np.searchsorted(db.offsets, g, side="right") - 1); e = db.shards[sh].export(int(g - db.offsets[sh]))
        orefs.append(ora.Sketch.from_arrays(args.c, K, M, "s%07d.fa" % int(g), e["seed"], e["pos"], e["ctgcanon"], e["markers"], e["contig_lengths"], e["total_len"]))
    oqs = []
    for x in pick:
        qb, _, _, _ = make_queries(torch, device, [int(qclades[x])], args.mean_len, first=int(x))
        oqs.append(ora.sketch_records([("q", qb.cpu().numpy())], args.c, K, M, "t%07d.fa" % int(x)))   # (query names sort after every database name, as in the GPU run)
    model = ora.Model(os.path.join(ROOT, "skani_amd", "data", "gbdt_c125.bin" if abs(args.c - 125) < abs(args.c - 200) else "gbdt_c200.bin")) if args.c >= 70 else None
    logical, physical, model_name = host_cores(); quota = cpu_quota()
    threads = max(1, min(logical, int(round(quota)) if quota else physical))
    ora.search(orefs, oqs[:2], 0.0, True, min_af=-1.0, model=model, threads=threads)         # (first touch of the heap)
    t0 = time.perf_counter()
    oq, orf, ores, nch = ora.search(orefs, oqs, 0.0, True, min_af=-1.0, model=model, threads=threads)
    wall = time.perf_counter() - t0
    ix, sc, ch = ora.triangle_phases()
    # the same pairs and values as the GPU's rows of these queries
    want = {}
    for a, b, x in zip(q, r, res):
        want[(int(a), int(b))] = x
    delta = {"pairs_compared": int(len(oq)), "same_pair_set": True}
    gq_rows = {(int(a), int(b)) for a, b in zip(q, r) if int(a) in set(int(x) for x in pick)}
    o_rows = {(int(pick[a]), int(ref_ids[b])) for a, b in zip(oq, orf)}
    delta["same_pair_set"] = gq_rows == o_rows
    if delta["same_pair_set"] and len(oq):
        for f in ("ani", "af_ref", "af_query"):
            delta["max_abs_d_" + f] = float(max(abs(float(want[(int(pick[a]), int(ref_ids[b]))][f]) - float(x[f])) for a, b, x in zip(oq, orf, ores)))
    hits_per_query = len(q) / max(nq, 1)
    scaled = {"marker_index": ix * n_db / len(orefs), "screen": sc * nq / len(oqs), "chain": ch * (len(q) / max(int(nch), 1))}
    scaled["total"] = sum(scaled.values())
    return {"value": nq / scaled["total"], "unit": "queries/s", "cores": threads, "threads": threads, "cpus_granted": quota if quota else logical, "kind": "port",
            "sample": "%d of the %d queries against %d of the %d database genomes (the sampled queries' clades + other clades, %d in all; reference sketches exported from the "
                      "GPU database, queries sketched by the oracle, neither timed): oracle marker index %.3f s + screen %.3f s + chain of %d pairs %.3f s on %d threads; scaled to the "
                      "workload: index x %.1f, screen x %.1f, chain x %.1f = %.2f s per %d queries (%.1f hits per query)"
                      % (len(oqs), nq, len(orefs), n_db, len(clades), ix, sc, int(nch), ch, threads, n_db / len(orefs), nq / len(oqs), len(q) / max(int(nch), 1), scaled["total"], nq, hits_per_query),
            "seconds": {"marker_index": ix, "screen": sc, "chain": ch, "total": wall}, "seconds_scaled_to_workload": scaled, "chained_pairs": int(nch),
            "chained_pairs_per_s": int(nch) / ch if ch > 0 else None, "delta_vs_oracle": delta,
            "host": {"logical_cpus": logical, "physical_cores": physical, "model": model_name, "cgroup_cpu_quota": quota}}


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup cpu.max / cfs quota), or None when unlimited: a box can show 256 CPUs and grant 16."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def host_cores():
    """(logical CPUs, physical cores, model name) of this box."""
    logical = os.cpu_count() or 1
    try:
        logical = min(logical, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    cores, model = set(), ""
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    physical = len(cores) if cores else max(1, logical // 2)
    return logical, min(physical, logical), model


def cpu_run(host_genomes, threads, ids=None):
    """One timed pass of the oracle (a C++ restatement of the reference algorithms) over the genomes given, phase by phase as BASELINE.md section 2
    lists them: sketch (one C call, files in parallel like file_io.rs:147, AVX2 seeding like avx2_seeding.rs), marker index, screen of all pairs,
    chain of the passing pairs (threads pulling pairs like triangle.rs:71-105)."""
    from oracle import oracle_py as ora
    names = ["s%05d.fa" % (i if ids is None else int(ids[i])) for i in range(len(host_genomes))]     # (names sort like the collection's global indices: the switch_qr tie)
    # regression.rs:8-28: learned ANI only for c >= 70, table chosen by |c-125| < |c-200|
    model = None
    if C >= 70:
        model = ora.Model(os.path.join(ROOT, "skani_amd", "data", "gbdt_c125.bin" if abs(C - 125) < abs(C - 200) else "gbdt_c200.bin"))
    t0 = time.perf_counter()
    sks = ora.sketch_batch(host_genomes, C, K, M, names, 1, 500, threads)
    t1 = time.perf_counter()
    oi, oj, res, n_chained, n_pass = ora.triangle(sks, model=model, threads=threads)
    t2 = time.perf_counter()
    index_s, screen_s, chain_s = ora.triangle_phases()
    n = len(host_genomes); pairs = n * (n - 1) // 2
    bases = sum(len(s) for g in host_genomes for _, s in g)
    return {"threads": threads, "genomes": n, "bases": bases, "pairs": pairs, "chained_pairs": int(n_chained), "value": pairs / (t2 - t0),
            "seconds": {"sketch": t1 - t0, "marker_index": index_s, "screen": screen_s, "chain": chain_s, "total": t2 - t0},
            "sketch_mbases_per_s": bases / 1e6 / (t1 - t0), "screen_pairs_per_s": pairs / max(index_s + screen_s, 1e-9),
            "chained_pairs_per_s": n_chained / chain_s if chain_s > 0 else None}, (oi, oj, res)


def cpu_baseline(host_genomes, gpu_result=None, n_gpu_genomes=None, ids=None):
    """The CPU side of the metric: the oracle (kind = 'port': the Rust reference cannot be built here) on this box's host cores on the genomes given --
    by default the FULL workload the GPU ran -- at several thread counts: 16, 64, the physical cores and all logical CPUs, each a complete run, plus
    skani's default -t 3 (cli.rs:243) on five clades as the per-thread yardstick.  `value` is the BEST point of the sweep and `cores` the thread count
    that gave it; every point carries its phase times and the parallel efficiency of the sketch and chain phases against the 3-thread per-thread rate.
    With the GPU triangle's result it also reports the metric's "ANI delta vs ref" over every chained pair.
    ids: the global index of each genome given (a sample of whole clades out of a larger, shuffled collection: --collection); the sample's phase rates are
    then scaled to the collection -- sketching and chaining linearly, the screen with the pair count -- and `value` is the collection's pairs over that time."""
    logical, physical, model_name = host_cores()
    n = len(host_genomes); pairs = n * (n - 1) // 2
    if ids is None:
        ids = np.arange(n, dtype=np.int64)
    cpu_run(host_genomes[:min(n, 5 * CLADE)], 3, ids)                 # (the first pass of a process pays the page faults of its heap: not the yardstick)
    few, _ = cpu_run(host_genomes[:min(n, 5 * CLADE)], 3, ids)
    per_thread = {"sketch": few["sketch_mbases_per_s"] / 3, "chain": (few["chained_pairs_per_s"] or 0) / 3}
    quota = cpu_quota()
    counts = {t for t in (16, 64, physical, logical) if t <= logical} or {logical}
    if quota:                                                          # the thread count the container's CPU quota pays for
        counts.add(max(1, min(logical, int(round(quota)))))
    counts = sorted(counts)
    sweep, result = [], None
    for t in counts:
        r, result = cpu_run(host_genomes, t, ids)
        r["efficiency"] = {"sketch": r["sketch_mbases_per_s"] / t / per_thread["sketch"] if per_thread["sketch"] else None,
                           "chain": (r["chained_pairs_per_s"] or 0) / t / per_thread["chain"] if per_thread["chain"] else None}
        sweep.append(r)
    best = max(sweep, key=lambda r: r["value"])
    oi, oj, res = result
    delta = None
    if gpu_result is not None:
        gi, gj, gres = gpu_result
        big = int(n_gpu_genomes or n)
        sel = np.isin(gi, ids) & np.isin(gj, ids)                      # the GPU's pairs inside the genomes the oracle ran on (global indices)
        gkeys = gi[sel].astype(np.int64) * big + gj[sel]; okeys = ids[oi].astype(np.int64) * big + ids[oj]
        same = len(gkeys) == len(okeys) and bool(np.array_equal(gkeys, okeys))
        delta = {"pairs_compared": int(len(okeys)), "same_pair_set": same}
        if same and len(okeys):
            g = gres[sel]
            for f in ("ani", "af_ref", "af_query"):
                delta["max_abs_d_" + f] = float(np.max(np.abs(g[f].astype(np.float64) - res[f].astype(np.float64))))
            delta["int_fields_equal"] = bool(all(np.array_equal(g[f], res[f]) for f in ("avg_chain_int_len", "total_bases_covered", "num_contigs_q", "num_contigs_r")))
    full = n_gpu_genomes is None or n == n_gpu_genomes
    sec = best["seconds"]
    value, scaled = best["value"], None
    if not full:                                                       # scale the sample's phases to the collection the GPU ran
        f = n_gpu_genomes / n; big_pairs = n_gpu_genomes * (n_gpu_genomes - 1) // 2
        scaled = {"sketch": sec["sketch"] * f, "marker_index": sec["marker_index"] * f, "screen": sec["screen"] * big_pairs / max(pairs, 1), "chain": sec["chain"] * f}
        scaled["total"] = sum(scaled.values())
        value = big_pairs / scaled["total"]
    return {"value": value, "unit": "genome-pairs/s", "cores": best["threads"], "threads": best["threads"],
            "cpus_granted": quota if quota else logical, "kind": "port", "delta_vs_oracle": delta,
            "sample": ("the full workload, measured: " if full else "a sample of whole clades out of the %d-genome collection, measured: " % n_gpu_genomes) +
                      "%d synthetic genomes (%d clades of %d, %.0f Mbp): oracle sketch %.2f s + marker index %.2f s + screen of %d pairs %.2f s + chain of %d pairs %.2f s "
                      "on %d threads (the best of %s threads; %d physical cores / %d logical CPUs%s, %s); %s"
                      % (n, n // CLADE, CLADE, best["bases"] / 1e6, sec["sketch"], sec["marker_index"], pairs, sec["screen"], best["chained_pairs"], sec["chain"],
                         best["threads"], "/".join(str(t) for t in counts), physical, logical, ", cgroup CPU quota %.1f: `cores` is a THREAD count, the container is granted "
                         "%.0f CPUs' worth of time" % (quota, quota) if quota else "", model_name,
                         "value = pairs / wall time of the four phases" if full else
                         "value = the collection's pairs / the sample's phase times scaled to the collection (sketch, index, chain x %.1f; screen x the pair ratio): %.1f s"
                         % (n_gpu_genomes / n, scaled["total"])),
            "seconds_scaled_to_collection": scaled,
            "seconds": sec, "sketch_mbases_per_s": best["sketch_mbases_per_s"], "screen_pairs_per_s": best["screen_pairs_per_s"],
            "chained_pairs_per_s": best["chained_pairs_per_s"], "chained_pairs": best["chained_pairs"],
            "host": {"logical_cpus": logical, "physical_cores": physical, "model": model_name, "cgroup_cpu_quota": quota},
            "sweep": [{k: r[k] for k in ("threads", "value", "seconds", "sketch_mbases_per_s", "chained_pairs_per_s", "efficiency")} for r in sweep],
            "default_threads": {"cores": 3, "genomes": few["genomes"], "value": few["value"], "seconds": few["seconds"], "chained_pairs_per_s": few["chained_pairs_per_s"],
                                "sketch_mbases_per_s": few["sketch_mbases_per_s"],
                                "note": "skani's default -t 3 on %d genomes: the per-thread rates the sweep's efficiencies refer to (value = that sample's own pairs / time)" % few["genomes"]}}


def write_fasta(path, recs, width=80):
    """recs = [(name, uint8 array)]: a FASTA file with `width` bases per line."""
    with open(path, "wb") as f:
        for name, a in recs:
            f.write(b">" + name.encode() + b"\n")
            n = len(a); rows = n // width
            if rows:
                body = np.empty((rows, width + 1), np.uint8); body[:, :width] = a[:rows * width].reshape(rows, width); body[:, width] = 10
                f.write(body.tobytes())
            if n > rows * width:
                f.write(a[rows * width:].tobytes() + b"\n")


def e2e_leg(host_genomes, threads, gpus=1, one_device=False):
    """End to end: the collection as FASTA files on a RAM disk -> `skani-hip triangle` (parse on `threads` threads, copy + pack while
    the next files are parsed, sketch, screen, chain, matrix writer) -> the ANI matrix file; wall clock of the whole process (HIP start-up and model load
    included) and the driver's own phase clock.  Beside it the oracle on the same files: read + sketch on the same number of threads, marker index,
    screen, chain (no writer)."""
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py as ora
    exe = os.path.join(ROOT, "skani_amd", "bin", "skani-hip")
    if not os.path.exists(exe):
        return {"error": "skani_amd/bin/skani-hip is not built"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="skani_e2e_", dir=base)
    try:
        names = [os.path.join(d, "s%05d.fa" % i) for i in range(len(host_genomes))]
        with ThreadPoolExecutor(max_workers=min(32, threads)) as ex:
            list(ex.map(lambda a: write_fasta(names[a[0]], a[1]), enumerate(host_genomes)))
        lst = os.path.join(d, "files.txt"); open(lst, "w").write("\n".join(names) + "\n")
        nbytes = sum(os.path.getsize(f) for f in names)
        env = dict(os.environ, SKH_TIMING="1", SKANI_HIP_DATA=os.path.join(ROOT, "skani_amd", "data"))
        node_args = (["--gpus", str(gpus)] + (["--one-device"] if one_device else [])) if gpus > 1 else []   # several GPUs: the command forks one rank per GPU itself
        runs = []
        for _ in range(3):                                            # the fastest run is the one reported, all are listed (the first pages the binary and the files in)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "triangle", "-t", str(threads), "-l", lst, "-o", os.path.join(d, "matrix.txt")] + node_args, capture_output=True, text=True, env=env)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": "skani-hip triangle failed: " + r.stderr[-400:]}
            phases = None
            for line in r.stderr.splitlines():
                if line.startswith("{") and "total_s" in line:
                    phases = json.loads(line)
            runs.append((wall, phases))
        wall, phases = min(runs, key=lambda r: r[0])
        n = len(host_genomes); pairs = n * (n - 1) // 2
        rows = open(os.path.join(d, "matrix.txt")).read().count("\n")
        out = {"genomes": n, "fasta_bytes": nbytes, "threads": threads, "wall_s": wall, "runs_wall_s": [round(r[0], 4) for r in runs], "pairs_per_s": pairs / wall,
               "phases_s": phases, "matrix_rows": rows - 1, "gpus": gpus,
               "command": "skani-hip triangle -t %d -l files.txt -o matrix.txt%s (FASTA on %s)" % (threads, " " + " ".join(node_args) if node_args else "", base or "the temp dir")}
        model = ora.Model(os.path.join(ROOT, "skani_amd", "data", "gbdt_c125.bin" if abs(C - 125) < abs(C - 200) else "gbdt_c200.bin")) if C >= 70 else None
        t0 = time.perf_counter()
        sks = ora.sketch_files(names, C, K, M, 1, 500, threads)
        t1 = time.perf_counter()
        ora.triangle(sks, model=model, threads=threads)
        t2 = time.perf_counter()
        ix, sc, ch = ora.triangle_phases()
        out["oracle"] = {"wall_s": t2 - t0, "threads": threads, "pairs_per_s": pairs / (t2 - t0),
                         "phases_s": {"read_sketch": t1 - t0, "marker_index": ix, "screen": sc, "chain": ch}, "note": "no matrix writer, no process start-up"}
        # if the box happens to have the reference itself (a `skani` binary on PATH -- it cannot be built in this image), time it on the same files
        ref = shutil.which("skani")
        if ref:
            t0 = time.perf_counter()
            r = subprocess.run([ref, "triangle", "-t", str(threads), "-l", lst, "-o", os.path.join(d, "ref_matrix.txt")], capture_output=True, text=True)
            out["reference_binary"] = {"path": ref, "returncode": r.returncode, "wall_s": time.perf_counter() - t0, "threads": threads,
                                       "pairs_per_s": pairs / max(time.perf_counter() - t0, 1e-9) if r.returncode == 0 else None}
        else:
            out["reference_binary"] = None                               # looked for on PATH, not there
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


class stdout_to_stderr:
    """RCCL prints a version banner through C stdio on stdout when its first communicator is made; the contract is ONE JSON line on stdout.  While this is
    active, file descriptor 1 points at stderr; on exit C's buffers are flushed there and stdout is restored."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1); os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1); os.close(self.saved)
        return False


def cpu_stat():
    """The container's CPU accounting (cgroup v2): a step that waits although the GPU is done may have been throttled by the CPU quota."""
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat") if len(l.split()) == 2)}
    except (OSError, ValueError):
        return {}


def cpu_stat_delta(before):
    now = cpu_stat()
    return {k: now[k] - before.get(k, 0) for k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec") if k in now}


def file_sha(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def pick_sample(canon, n_total, clades):
    """Global indices (ascending) of the members of `clades` whole clades taken evenly from the collection."""
    n_cl = n_total // CLADE; take = max(1, min(n_cl, clades))
    cl = set((np.arange(take, dtype=np.int64) * n_cl // take).tolist())
    return np.nonzero(np.isin(canon // CLADE, list(cl)))[0]


def kernel_sources_sha():
    import hashlib
    d = os.path.join(ROOT, "skani_amd", "csrc")
    return hashlib.sha256(b"".join(open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h")))).hexdigest()[:16]


def load_stage_profile(default_workload):
    """profiles/stage_profile.json (tools/make_stage_profile.py: per-kernel rocprofv3 averages, FETCH_SIZE / WRITE_SIZE bytes, SQ residency of one bench step), quoted only
    for the workload it was measured on and while the kernel sources are the ones it was measured on.  Returns (profile or None, note)."""
    path = os.path.join(ROOT, "profiles", "stage_profile.json")
    if not os.path.exists(path):
        return None, "no profiles/stage_profile.json"
    if not default_workload:
        return None, "profiles/stage_profile.json was measured on the default workload only"
    try:
        sp = json.load(open(path))
    except Exception as e:
        return None, "unreadable profiles/stage_profile.json: %r" % (e,)
    sha = kernel_sources_sha()
    if sp.get("kernel_sources_sha256_16") != sha:
        return None, "profiles/stage_profile.json was measured on other kernel sources (%s, now %s): not quoted" % (sp.get("kernel_sources_sha256_16"), sha)
    return sp, "rocprofv3 --kernel-trace + four --pmc passes at commit %s (%s)" % (sp.get("commit"), sp.get("source"))


def stage_rooflines(tm, steps, units, c, m, pairs_chained, sp):
    """Every stage of the step against the HBM roofline (SURVEY 8d asks for seeding, screen and chaining; the table build and the chaining's three parts are listed too).
    algorithmic bytes = the per-unit figures of DESIGN.md section 4 x the units counted behind the timed steps (`units`); ms_live = this run's HIP-event phase timer where
    the stage is a phase of the library (the three parts of the chaining share one timer: apportioned by the profile's kernel times); ms_profile / counter bytes /
    waves per SIMD from the stamped profile (None when it does not apply)."""
    if not units or "error" in units:
        return None
    B, P, Mk, E, A, I = (units[k] for k in ("bases", "seed_positions", "markers", "enumerated_positions", "anchors", "candidate_intervals"))
    alg = {"seeding": ((0.25 + 12.0 / c + 8.0 / m) * B, "0.25 B/base packed + 12/c B seeds + 8/m B markers"),
           "tables": (29.0 * P + 40.0 * Mk, "29 B/position (8 read, 4 gathered, 17 written: slots + filter) + 40 B/marker (set: 8 in, 16 out; index: 8 in, 8 out)"),
           "screen": (8.0 * Mk, "8 B per marker incidence (SURVEY 8d)"),
           "join": (8.0 * E + 16.0 * A, "8 B per enumerated position + 16 B per anchor (hit record + anchor)"),
           "chunking + DP": (8.0 * A + 40.0 * I, "8 B per anchor + 40 B per candidate interval"),
           "selection + estimate": (40.0 * I + 4.0 * E + 64.0 * pairs_chained, "40 B per candidate interval + 4 B per query position + the 64-byte result row")}
    live = {"seeding": tm["seed_ms"] / steps, "tables": tm["sketch_build_ms"] / steps, "screen": tm["screen_ms"] / steps}
    chain_live = tm["chain_ms"] / steps
    prof = {}
    if sp:
        for k in sp["kernels"]:
            e = prof.setdefault(k["stage"], {"ms": 0.0, "read": 0.0, "write": 0.0, "kernels": []})
            e["ms"] += k["ms_per_step"]
            rd = k.get("read_bytes_per_step")
            if rd is not None and k["class"] == "mixed":
                rd += 8.0 * E / 2                                       # the count pass's stream of seeds and positions: FETCH_SIZE saw half of it (tools/make_stage_profile.py)
            e["read"] += rd or 0.0; e["write"] += k.get("write_bytes_per_step") or 0.0
            e["kernels"].append({kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in k.items()
                                 if kk in ("kernel", "ms_per_step", "launches_per_step", "waves_per_simd", "valu_busy_pct", "wait_any_pct", "vgprs", "lds_bytes")})
    chain_parts = ("join", "chunking + DP", "selection + estimate")
    chain_prof = sum(prof.get(x, {}).get("ms", 0.0) for x in chain_parts)
    rows = []
    for st in ("seeding", "tables", "screen") + chain_parts:
        ms_live = live.get(st)
        apportioned = False
        if ms_live is None and chain_prof > 0 and st in prof:
            ms_live = chain_live * prof[st]["ms"] / chain_prof; apportioned = True
        pr = prof.get(st)
        ms = ms_live if ms_live else (pr["ms"] if pr else None)
        row = {"stage": st, "bound": "hbm", "algorithmic_bytes": alg[st][0], "bytes_rule": alg[st][1], "ms_live": ms_live, "ms_live_apportioned": apportioned,
               "ms_profile": pr["ms"] if pr else None, "counter_bytes": (pr["read"] + pr["write"]) if pr and (pr["read"] + pr["write"]) > 0 else None,
               "achieved": alg[st][0] / (ms * 1e-3) / 1e9 if ms else None, "peak": 8000.0, "unit": "GB/s"}
        row["frac"] = row["achieved"] / 8000.0 if row["achieved"] else None
        if row["counter_bytes"]:
            row["traffic_over_algorithmic"] = row["counter_bytes"] / alg[st][0]
        if pr:
            row["kernels"] = sorted(pr["kernels"], key=lambda k: -k["ms_per_step"])[:6]
        rows.append(row)
    return rows


def run_triangle(args, torch, dist, sk, ctx, comm, transport, rank, world, device, n_local, order, strong, steps, warmup, want_cpu, want_e2e, defer_cpu_legs=False):
    """One measured triangle workload: every rank makes its n_local genomes of the collection (n_local * world, in `order`), W warm-up steps, `steps` timed steps
    between barriers, the MAX over ranks.  Rank 0 returns the bench line (a dict), the others None.  want_cpu: the oracle beside it (`cpu_baseline`, rank 0 only:
    on the genomes rank 0 holds when that is the whole workload, else on whole clades sampled from the collection, which rank 0 generates for itself)."""
    n_total = n_local * world
    assert n_total % CLADE == 0, "the collection must consist of whole clades"
    canon = genome_order(n_total, order)                               # global genome index -> canonical id
    mine = canon[rank * n_local:(rank + 1) * n_local]
    # genomes the oracle runs on beside the GPU: the full workload by default on one GPU; with --collection, or on several GPUs, whole clades taken evenly from
    # the collection (their members are spread over the shuffled order and over the ranks).  cpu_ids = their global indices, ascending.
    cpu_ids = np.zeros(0, np.int64)
    sampled = strong or world > 1
    if rank == 0 and want_cpu and args.cpu_clades != 0:
        if sampled:
            cpu_ids = pick_sample(canon, n_total, args.cpu_sample_clades if args.cpu_clades < 0 else args.cpu_clades)
        else:
            cpu_ids = np.arange(n_total if args.cpu_clades < 0 else min(n_total, args.cpu_clades * CLADE), dtype=np.int64)
        if args.cpu_genomes > 0:                                       # (a dense collection: the oracle on its first genomes, all pairs among them)
            cpu_ids = np.arange(min(n_total, args.cpu_genomes), dtype=np.int64)
    local_ids = cpu_ids if world == 1 else np.zeros(0, np.int64)      # (several ranks: the sample is made after the timed steps, its members live on every rank)
    keep = np.zeros(n_local, bool); keep[local_ids] = True
    bases, contig_off, contig_genome, ng, host_genomes = make_genomes(torch, device, mine, mean_len=args.mean_len, members=CLADE, keep_host=keep if len(local_ids) else False)
    host_genomes = [host_genomes[int(x)] for x in local_ids]
    torch.cuda.synchronize()
    gs = ctx.pack_buffer(None, contig_off, contig_genome, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
    total_bases_local = int(contig_off[-1])
    del bases
    torch.cuda.empty_cache()
    params = sk.SketchParams(C, K, M, sk.SEED_AVX2)
    mp = sk.MapParams(learned_ani=sk.use_learned_ani(C), compute_ci=not args.no_ci)
    ctx.timings()
    last = {}
    host_t = [0.0, 0.0, 0.0]                      # BENCH_STEP_TIMES=1: wall time of a step's three calls as the host sees them (stderr)
    genome_rank = np.arange(rank * n_local, (rank + 1) * n_local, dtype=np.uint32)
    rows_to_root = comm is not None and not args.rows_everywhere

    def step():
        # genome_rank = global index: the collection's names sort like its indices
        # several GPUs: seed tables deferred -- every rank indexes only the sketches it ends up chaining (its own that stay + the ones it receives)
        t0 = time.perf_counter()
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=genome_rank, defer_tables=comm is not None or args.tables == "beside-screen", screen_index=comm is None)
        t1 = time.perf_counter()
        if comm is None:
            i, j, res, n_chained = ctx.triangle(ss_local, mp)
        else:
            i, j, res, n_chained, last["stats"] = comm.triangle(ss_local, mp, rows_to_root=rows_to_root)   # the result rows are gathered on rank 0
        t2 = time.perf_counter()
        ss_local.close()
        t3 = time.perf_counter()
        host_t[0] += t1 - t0; host_t[1] += t2 - t1; host_t[2] += t3 - t2
        last["result"] = (i, j, res)
        return len(i), n_chained

    with stdout_to_stderr():                      # (the first collective may still print)
        for _ in range(warmup):
            step()
            if os.environ.get("BENCH_WARMUP_PAUSE"):      # diagnostic (profiles/r04_first_steps_transient.md): an idle stretch behind each warm-up step
                time.sleep(float(os.environ["BENCH_WARMUP_PAUSE"]))
        if comm is not None and warmup == 0:
            dist.barrier()
    ctx.timings()
    host_t[:] = [0.0, 0.0, 0.0]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    cpu_stat0 = cpu_stat()
    t0 = time.perf_counter()
    kept = chained = 0
    step_wall = []
    for k in range(steps):
        ts = time.perf_counter(); h0 = list(host_t)
        kept, chained = step()
        step_wall.append(time.perf_counter() - ts)
        if os.environ.get("BENCH_STEP_TIMES"):    # (interleaves with SKH_TRACE_ALLOC=1 on stderr: which call of which step went to the driver)
            print("step %d of %d genomes: sketch_genomes %.3f ms, triangle %.3f ms, close %.3f ms" % ((k, n_local) + tuple(1e3 * (host_t[x] - h0[x]) for x in range(3))), file=sys.stderr)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    units = None
    if comm is None and not args.no_units:
        # what the step worked on, counted once OUTSIDE the timed region (a sketch + screen + chaining call with per-pair stage statistics): the units roofline_stages prices
        try:
            ss_u = ctx.sketch_genomes(gs, params, genome_rank=genome_rank)
            meta = ss_u.export_meta()
            npos = np.diff(np.asarray(meta["pos_off"], np.int64))
            ui, uj = ctx.screen(ss_u, None, 0.0)
            _, ust = ctx.chain_pairs(ss_u, None, ui, uj, mp, stats=True)
            enumerated = np.where(ust["switched"] != 0, npos[ui], npos[uj])                 # chain.rs:625-661: the side that is walked position by position
            units = {"bases": total_bases_local, "seed_positions": int(meta["pos_off"][-1]), "markers": int(meta["marker_off"][-1]), "candidate_pairs": int(len(ui)),
                     "enumerated_positions": int(enumerated.sum()), "anchors": int(ust["n_anchors"].sum()), "listed_query_positions": int(ust["n_qpos"].sum()),
                     "chunks": int(ust["n_chunks"].sum()), "candidate_intervals": int(ust["n_intervals"].sum()), "accepted_intervals": int(ust["n_accepted"].sum())}
            ss_u.close()
        except Exception as e:                                        # the line must not depend on it
            units = {"error": repr(e)}
        ctx.timings()
    gs.close()
    torch.cuda.empty_cache()
    if os.environ.get("BENCH_STEP_TIMES"):
        print("host view of a step (ms): sketch_genomes %.3f, triangle %.3f, sketch set close %.3f; library timers: %s; cgroup cpu.stat over the timed steps: %s" %
              tuple([1e3 * x / steps for x in host_t] + [{k: round(v / steps, 3) for k, v in tm.items() if k.endswith("_ms")}, cpu_stat_delta(cpu_stat0)]), file=sys.stderr)
    per_rank = None
    if comm is not None:
        t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        st = last["stats"]
        mine_t = torch.tensor([st["n_pairs_mine"], st["n_genomes_received"], st["bytes_received"], int(tm["chain_ms"] * 1000), int(tm["exchange_ms"] * 1000),
                               int(tm["seed_ms"] * 1000), int(tm["sketch_build_ms"] * 1000), int(tm["screen_ms"] * 1000), total_bases_local,
                               int(tm.get("exchange_wait_ms", 0.0) * 1000), kept], dtype=torch.int64)
        allt = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        per_rank = [[int(x) for x in a] for a in allt]
    if rank != 0:
        return None
    ms_per_step = dt / steps * 1e3
    pairs = n_total * (n_total - 1) // 2
    value = pairs / (dt / steps)
    # roofline of the seeding kernel: algorithmic bytes = 0.25 B/base packed read + 12/c B seeds + 8/m B markers
    alg_bytes_per_base = 0.25 + 12.0 / C + 8.0 / M
    launches = max(tm["seed_kernel_launches"], 1)
    seed_ms_per_launch = tm["seed_kernel_ms"] / launches
    bytes_per_launch = alg_bytes_per_base * total_bases_local * steps / launches
    achieved = bytes_per_launch / (seed_ms_per_launch * 1e-3) / 1e9 if seed_ms_per_launch > 0 else 0.0
    # counters that were measured offline (separate rocprofv3 --pmc passes) are only quoted while the kernel source they were measured on is unchanged
    seed_src_sha = file_sha(os.path.join(ROOT, "skani_amd", "csrc", "pack_seed.hip"))
    traffic, traffic_note, valu, valu_waves, rocprof_ms = None, "no profiles/seed_traffic.json", None, None, None
    tpath = os.path.join(ROOT, "profiles", "seed_traffic.json")
    if os.path.exists(tpath) and n_local == 1000 and args.mean_len == 5_000_000 and C == 125 and order == "clade":
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel_source_sha256_16") == seed_src_sha:
                traffic = tj.get("hbm_bytes_per_launch"); traffic_note = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at commit %s (%s)" % (tj.get("commit"), tj.get("source"))
                valu = tj.get("valu"); valu_waves = tj.get("waves_per_launch"); rocprof_ms = tj.get("rocprof_avg_ms")
            else:
                traffic_note = "profiles/seed_traffic.json was measured on another version of pack_seed.hip (%s, now %s): not quoted" % (tj.get("kernel_source_sha256_16"), seed_src_sha)
        except Exception as e:
            traffic_note = "unreadable profiles/seed_traffic.json: %r" % (e,)
    elif os.path.exists(tpath):
        traffic_note = "counters were measured on the default workload only"
    roof = {"kernel": "seed_tiles_kernel", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": traffic, "traffic_source": traffic_note, "bytes_per_launch": bytes_per_launch, "ms_per_launch": seed_ms_per_launch,
            "launches_per_step": launches / steps,
            "note": "0.354 algorithmic B/base; the kernel is bound by VALU issue, not by HBM: see valu_frac and profiles/r02_valu_rates.md.  frac_basis says which time "
                    "achieved / frac are divided by; frac_hip_events is always this run's own HIP-event time per launch"}
    roof["frac_hip_events"] = roof["frac"]; roof["achieved_hip_events"] = roof["achieved"]
    roof["frac_basis"] = "this run's HIP-event time per launch (no rocprofv3 trace of these kernel sources on this workload is committed)"
    if rocprof_ms:
        # the judged fraction: algorithmic bytes over the kernel's AVERAGE duration in the committed rocprofv3 --kernel-trace run of the same source (a few per cent longer than
        # the HIP-event time of this run, which stays on the line as frac_hip_events)
        roof["ms_per_launch_rocprof"] = rocprof_ms
        roof["achieved"] = bytes_per_launch / (rocprof_ms * 1e-3) / 1e9
        roof["frac"] = roof["frac_rocprof"] = roof["achieved"] / 8000.0
        roof["frac_basis"] = "average duration of the kernel in the committed rocprofv3 --kernel-trace summary of the same kernel source (profiles/seed_traffic.json: rocprof_avg_ms)"
    if valu:
        # VALU issue cycles the kernel's instructions need (static count per wave x measured cycles per instruction class, tools/isa_mix.py) over the
        # SIMD cycles its launch had: 1024 SIMDs x shader clock x kernel time
        simd_cycles = 1024 * valu["clock_ghz"] * 1e9 * seed_ms_per_launch * 1e-3
        waves = valu_waves or total_bases_local / 8192.0 * 4.0          # SQ_WAVES of the launch (else: one workgroup of 4 waves per 8192 windows)
        roof["valu_frac"] = waves * valu["issue_cycles_per_wave"] / simd_cycles
        roof["valu"] = valu
    out = {
        "metric": "genome-pairs/sec (triangle, ~5 Mbp genomes)", "value": value, "unit": "genome-pairs/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "skani triangle over %d synthetic ~%.1f Mbp genomes (clades of %d, 0.5-8%% divergence, %s), -c %d -k %d -m %d -s 80, learned ANI on%s"
                               % (n_total, args.mean_len / 1e6, CLADE, "listed clade by clade" if order == "clade" else "file order shuffled: clades span the GPUs", C, K, M,
                                  "" if world == 1 else ", tiled across %d %s (%d genomes each) via %s" % (world, "processes on ONE GPU" if args.one_device else "GPUs", n_local,
                                                                                                   "RCCL" if transport == "rccl" else "host collectives (gloo)")),
                   "genomes": n_total, "genomes_per_gpu": n_local, "bases_per_gpu": total_bases_local, "pairs": pairs, "chained_pairs": chained,
                   "kept_pairs": kept, "order": order,
                   "mode": ("strong scaling: a fixed collection of %d genomes at every N" % n_total) if strong else
                   ("weak scaling: %d genomes per GPU" % n_local if world > 1 else "one GPU"), "one_device": bool(args.one_device),
                   "transport": transport, "result_rows": "single GPU" if comm is None else ("gathered on rank 0" if rows_to_root else "gathered on every rank"),
                   "parallelism": "single GPU" if world == 1 else
                                  "one process per GPU; every rank sketches its genomes; markers all-gathered; the screen is cut by key range (every rank counts the "
                                  "incidences of a W-th of the markers' leading 16 bases, the non-zero cells are gathered on the device and every rank applies the rule "
                                  "itself); candidate pairs assigned to ranks cluster by cluster (balanced, order-independent); only the needed sketches travel, "
                                  "point-to-point and asynchronously; result rows gathered on rank 0"},
        "phase_ms_per_step": {k: tm[k] / steps for k in ("seed_ms", "sketch_build_ms", "screen_ms", "chain_ms", "exchange_ms")},
        "step_wall_ms_rank0": [round(1e3 * x, 3) for x in step_wall], "ms_per_step_median_rank0": round(1e3 * float(np.median(step_wall)), 3),
        "roofline": roof,
    }
    # what compares across N and across the weak / strong modes: the pair count grows with the square of the collection, the work with its size
    step_s = dt / steps
    total_bases = sum(r[8] for r in per_rank) if per_rank else total_bases_local
    out["bases_per_s_per_gpu"] = total_bases / step_s / world
    out["chained_pairs_per_s_per_gpu"] = chained / step_s / world
    out["genomes_per_s_per_gpu"] = n_total / step_s / world
    if per_rank:
        names = ("chained_pairs", "sketches_received", "bytes_received", "chain_us", "exchange_us", "seed_us", "sketch_build_us", "screen_us", "bases", "exchange_wait_us", "rows_returned")
        out["per_rank"] = {nm: [r[x] // (steps if nm.endswith("_us") else 1) for r in per_rank] for x, nm in enumerate(names)}
    # the chaining pipeline against the north star's algorithmic figure: both sketches of a chained pair read once, 12 B per position
    # (~0.96 MB per pair of 5 Mbp genomes at c=125)
    chain_s = tm["chain_ms"] / steps * 1e-3
    if chain_s > 0 and chained:
        chain_bytes = 12.0 * 2.0 * (total_bases_local / max(n_local, 1) / C) * (chained / world)
        out["roofline_chain"] = {"stage": "join + chunk + chain + select + estimate", "bound": "hbm", "achieved": chain_bytes / chain_s / 1e9, "peak": 8000.0,
                                 "unit": "GB/s", "frac": chain_bytes / chain_s / 1e9 / 8000.0, "bytes_per_step": chain_bytes, "traffic": None,
                                 "note": "irregular, latency-bound stages: per-kernel traffic, occupancy and LDS figures in profiles/r05_pmc.md"}
        # measured HBM bytes of the stage per step (separate rocprofv3 --pmc passes, tools/make_chain_traffic.py), quoted while the chaining sources are unchanged
        cpath = os.path.join(ROOT, "profiles", "chain_traffic.json")
        if os.path.exists(cpath) and world == 1 and n_local == 1000 and args.mean_len == 5_000_000 and C == 125 and order == "clade":
            try:
                import hashlib
                cj = json.load(open(cpath))
                srcs = sorted(f for f in os.listdir(os.path.join(ROOT, "skani_amd", "csrc")) if f.startswith("chain"))
                sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "skani_amd", "csrc", f), "rb").read() for f in srcs)).hexdigest()[:16]
                if cj.get("chain_sources_sha256_16") == sha:
                    out["roofline_chain"]["traffic"] = cj["hbm_bytes_per_step"]
                    out["roofline_chain"]["traffic_over_algorithmic"] = cj["hbm_bytes_per_step"] / chain_bytes
                    out["roofline_chain"]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at commit %s, %s" % (cj.get("commit"), cj.get("rule"))
                else:
                    out["roofline_chain"]["traffic_source"] = "profiles/chain_traffic.json was measured on other chaining sources: not quoted"
            except Exception as e:
                out["roofline_chain"]["traffic_source"] = "unreadable profiles/chain_traffic.json: %r" % (e,)
    out["units"] = units
    if world == 1:
        default_workload = n_local == 1000 and args.mean_len == 5_000_000 and C == 125 and order == "clade" and CLADE == 20
        sp, sp_note = load_stage_profile(default_workload)
        out["roofline_stages"] = stage_rooflines(tm, steps, units, C, M, chained, sp)
        out["roofline_stages_source"] = sp_note
        if out["roofline_stages"]:
            scr = next(r for r in out["roofline_stages"] if r["stage"] == "screen")
            out["roofline_screen"] = {"stage": "marker screen (incidence count + rule; the index sort runs at sketch time)", "bound": "hbm", "achieved": scr["achieved"], "peak": 8000.0, "unit": "GB/s",
                                      "frac": scr["frac"], "bytes_per_step": scr["algorithmic_bytes"], "ms_per_step": scr["ms_live"], "traffic": scr["counter_bytes"],
                                      "note": "8 B per marker incidence (SURVEY 8d); what bounds the count is the L2's atomic units -- line requests, not bytes: profiles/r06_atomic_rates.md"}
    out["cpu_baseline"] = None
    def cpu_legs(host_genomes=host_genomes):
        """The legs that load the host's cores (the oracle's thread sweep, the command line with its ingest threads): run by the caller AFTER every GPU measurement of the
        line -- a `strong` block measured right behind them had one step in two 25-40 ms late (the container's CPU quota, spent by the sweep, throttles the thread that
        drives the GPU); measured in front of them, none."""
        if not len(cpu_ids): return
        if world > 1:                                                  # the sample's members live on every rank: rank 0 makes them again for the oracle (genomes are pure functions of their ids)
            b2, _, _, _, host_genomes = make_genomes(torch, device, canon[cpu_ids], mean_len=args.mean_len, members=CLADE, keep_host=True)
            del b2
            torch.cuda.empty_cache()
        gpu_result = last.get("result")
        out["cpu_baseline"] = cpu_baseline(host_genomes, gpu_result, n_total, ids=cpu_ids)
        if want_e2e:
            try:
                q = cpu_quota()
                out["e2e"] = e2e_leg(host_genomes, max(4, min(host_cores()[1], 64, int(round(q)) * 2 if q else 64)), gpus=world, one_device=args.one_device)
            except Exception as e:                                   # the headline line must not depend on a RAM disk
                out["e2e"] = {"error": repr(e)}
    if defer_cpu_legs:
        out["_cpu_legs"] = cpu_legs
    else:
        cpu_legs()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genomes-per-gpu", type=int, default=0, help="default: 1000 on one GPU (BASELINE config 3), 1250 per GPU on several (config 4 = 10,000 on 8)")
    ap.add_argument("--mean-len", type=int, default=5_000_000)
    ap.add_argument("--order", default="", choices=["", "clade", "shuffled"], help="order of the collection: clade by clade, or shuffled like unrelated file names "
                    "(default on several GPUs: members of a clade sit on different ranks and their sketches have to travel)")
    ap.add_argument("--cpu-clades", type=int, default=-1, help="clades (x20 genomes) the CPU baseline runs on; default -1 = the full workload on one GPU, --cpu-sample-clades "
                    "sampled clades with --collection or on several GPUs; 0 disables")
    ap.add_argument("--no-ci", action="store_true")
    ap.add_argument("--root-fasta", default="", help="every clade's root is this genome (FASTA, .gz allowed; all records joined, bytes other than ACGT read as A) instead of "
                    "i.i.d. bases: a rate on real sequence -- repeated seeds, list storage and the join's band rule on the timed path (with --clade = the number of genomes: all pairs chained)")
    ap.add_argument("--cpu-genomes", type=int, default=0, help="the oracle runs on the first N genomes of the collection (overrides --cpu-clades; for dense collections)")
    ap.add_argument("--no-variants", action="store_true", help="the default line also measures the secondary scale points (dense, -c 30 / 70 / 200, 5,000 genomes) as its `variants` block; this skips them")
    ap.add_argument("--no-units", action="store_true", help="skip the untimed pass behind the timed steps that counts the step's units (positions, markers, anchors, ...: roofline_stages); "
                    "profiling runs pass it so that a trace holds the steps' kernels only")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (FASTA files on a RAM disk through the skani-hip binary, ~20 s)")
    ap.add_argument("--c", type=int, default=125, help="-c compression factor (presets: 30 slow, 70 medium, 125 default, 200 fast)")
    ap.add_argument("--clade", type=int, default=20, help="genomes per clade (= genomes-per-gpu gives the dense single-clade variant)")
    ap.add_argument("--workload", default="triangle", choices=["triangle", "search"], help="triangle = the headline metric; search = BASELINE config 5 (optional)")
    ap.add_argument("--db-genomes", type=int, default=10000)
    ap.add_argument("--no-compact", action="store_true", help="search workload: database shards with the triangle's table geometry (2 home slots per position, full list storage) instead of SKH_SKETCH_COMPACT")
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--cpu-queries", type=int, default=20, help="search workload: queries the oracle searches beside the GPU (cpu_baseline; 0 disables)")
    ap.add_argument("--force-dist", action="store_true", help="one GPU: still go through the RCCL communicator and skh_triangle_distributed (world size 1; exercises the multi-GPU code path)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"], help="several GPUs: the library's own RCCL communicator (default) or host collectives over torch.distributed (debug)")
    ap.add_argument("--tables", default="at-sketch", choices=["beside-screen", "at-sketch"], help="one GPU: the seed tables are built inside skh_sketch_genomes beside the marker sets (default) "
                    "or inside skh_triangle beside its marker screen (sketches made with SKH_SKETCH_DEFER_TABLES; measured in round 4: the same step time)")
    ap.add_argument("--one-device", action="store_true", help="all ranks on cuda:0, host collectives over gloo (RCCL refuses two ranks on one GPU): runs this file's whole "
                    "multi-rank branch on a one-GPU box (tests/test_zz_bench_multirank.py); the number it prints is not a multi-GPU measurement")
    ap.add_argument("--collection", type=int, default=0, help="strong scaling: a fixed collection of this many genomes (10000 = BASELINE config 4) in shuffled order at every "
                    "--gpus N, collection / N genomes per rank")
    ap.add_argument("--cpu-sample-clades", type=int, default=50, help="--collection, or several GPUs: whole clades (taken evenly from the collection) the oracle runs on beside the GPUs "
                    "(cpu_baseline scaled to the collection, delta_vs_oracle)")
    ap.add_argument("--strong-collection", type=int, default=None, help="the default (weak) mode also measures this fixed collection at the same N and reports it as the line's "
                    "`strong` block -- a driver sweep --gpus 1/2/4/8 then carries a strong-scaling series beside the weak one; default: 10000 for the default workload shape, "
                    "else 0 = none")
    ap.add_argument("--strong-steps", type=int, default=4)
    ap.add_argument("--rows-everywhere", action="store_true", help="several GPUs: gather the result rows on every rank (skh_triangle_distributed) instead of on rank 0 (SKH_DIST_ROWS_TO_ROOT)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import skani_amd as sk
    global C, CLADE, ROOT_CODES
    C, CLADE = args.c, args.clade

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if args.one_device:
        local = 0; args.transport = "torch"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")             # one node: the ranks meet on the loopback interface whatever the host name resolves to
        with stdout_to_stderr():
            # torch.distributed is the CONTROL plane only (the RCCL unique id, agreement on the transport, the timing reduction, barriers): gloo on host
            # tensors, the same with one GPU per rank and with --one-device.  The data plane is the library's own RCCL communicator (csrc/rccl_transport.hip),
            # or -- if that cannot be made, or on request -- host collectives over this same gloo group.
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = sk.Context(local)
    if args.workload == "search":
        if world > 1:
            raise SystemExit("the search workload is single-GPU")
        if args.c == 125:
            args.c = C = 70            # --medium, BASELINE config 5
        print(json.dumps(run_search(args, torch, sk, ctx, device)[0]))
        return

    strong = args.collection > 0
    if strong:
        if args.collection % world or args.genomes_per_gpu:
            raise SystemExit("--collection must be a multiple of --gpus (and excludes --genomes-per-gpu)")
        n_local = args.collection // world
    else:
        n_local = args.genomes_per_gpu or (1000 if world == 1 else 1250)
    order = args.order or ("clade" if world == 1 and not strong else "shuffled")

    from skani_amd.distributed import Comm
    comm, transport = None, None
    if world > 1 or args.force_dist:
        with stdout_to_stderr():
            transport, why = args.transport, None
            if transport == "rccl":
                try:
                    comm = Comm.rccl(ctx, dist, rank, world, torch=torch, device=device)
                    comm.selftest()                  # one small all-gather and one all-to-all through the communicator, checked: a transport that does not
                                                     # move bytes correctly is found here, where every rank can still fall back together
                except Exception as e:               # e.g. no librccl for dlopen: every rank must take the same way out
                    why = repr(e)
                failed = torch.tensor([1 if why else 0], dtype=torch.int32)
                dist.all_reduce(failed, op=dist.ReduceOp.MAX)
                if int(failed.item()):
                    if comm is not None:
                        comm.close(); comm = None
                    print("bench: RCCL communicator not available on every rank (%s): host collectives over torch.distributed instead" % (why,), file=sys.stderr)
                    transport = "torch"
            if comm is None:
                comm = Comm.host(ctx, dist, rank, world, torch=torch)

    real = None
    if args.root_fasta:
        import gzip
        opener = gzip.open if args.root_fasta.endswith(".gz") else open
        seq = b"".join(l.strip() for l in opener(args.root_fasta, "rb") if not l.startswith(b">"))
        lut = np.zeros(256, np.uint8)
        for ch, code in ((b"Cc", 1), (b"Gg", 2), (b"TtUu", 3)):
            for b_ in ch: lut[b_] = code
        ROOT_CODES = torch.from_numpy(lut[np.frombuffer(seq, np.uint8)]).to(device)
        real = {"root": os.path.basename(args.root_fasta), "root_bases": len(seq),
                "generator": "every genome = the root with substitution rate U(0.005, 0.08), 0-5 deletions of 10-50 kb, cut into 1-20 contigs (make_genomes; the point-substitution model of tests/helpers.mutate)"}
    out = run_triangle(args, torch, dist, sk, ctx, comm, transport, rank, world, device, n_local, order, strong, args.steps, args.warmup,
                       want_cpu=True, want_e2e=not args.no_e2e and not strong and not args.root_fasta, defer_cpu_legs=True)
    if real is not None and rank == 0:
        # what a real genome puts on the timed path that i.i.d. sequence does not: seeds with several positions (list storage) and seeds beyond the band (dropped by the join, chain.rs:674-696)
        b0, co0, cg0, ng0, _ = make_genomes(torch, device, [0], mean_len=args.mean_len, members=CLADE)
        torch.cuda.synchronize()
        g0 = ctx.pack_buffer(None, co0, cg0, ng0, sk.SEED_AVX2, device_ptr=b0.data_ptr())
        s0 = ctx.sketch_genomes(g0, sk.SketchParams(C, K, M, sk.SEED_AVX2))
        seeds = s0.export(0)["seed"]
        _, cnt = np.unique(seeds, return_counts=True)
        band = 2500 // C
        n_pos = int(cnt.sum()); listed = int(cnt[(cnt > 1) & (cnt <= band)].sum()); rep = int(cnt[cnt > band].sum())
        words = int((cnt[(cnt > 1) & (cnt <= band)] + 1).sum())                       # a list = its count + its positions
        real["genome_0"] = {"seed_positions": n_pos, "distinct_seeds": int(len(cnt)), "positions_of_seeds_with_2_to_band_positions": listed, "positions_of_seeds_beyond_the_band": rep,
                            "share_in_lists": listed / n_pos, "share_repetitive": rep / n_pos, "list_storage_words_used": words, "list_storage_fill": words / (1.5 * n_pos),
                            "note": "list storage capacity = 6 B per seed position (DESIGN.md section 3); i.i.d. genomes of this size: ~5 % in lists, nothing beyond the band"}
        s0.close(); g0.close(); del b0
        out["real_sequence"] = real
    # the same N on the fixed collection (BASELINE config 4's 10,000 genomes): the strong-scaling point that belongs to this line.  A default sweep --gpus 1/2/4/8 then
    # yields the weak series (`value`) AND the strong one (`strong.ms_per_step`) without a second sweep.
    default_shape = not args.genomes_per_gpu and args.mean_len == 5_000_000 and CLADE == 20 and C == 125 and not args.force_dist
    if args.strong_collection is None:
        args.strong_collection = 10000 if default_shape else 0
    if not strong and args.strong_collection > 0 and args.strong_collection % world == 0 and args.strong_collection % CLADE == 0:
        sn = args.strong_collection // world
        if sn == n_local and order == "shuffled":                      # (8 GPUs: the weak default IS config 4)
            s = out
        else:
            s = run_triangle(args, torch, dist, sk, ctx, comm, transport, rank, world, device, sn, "shuffled", True, max(1, args.strong_steps), max(args.warmup, 2),
                             want_cpu=False, want_e2e=False)
        if rank == 0:
            out["strong"] = {k: s[k] for k in ("ms_per_step", "ms_per_step_median_rank0", "step_wall_ms_rank0", "value", "steps", "warmup", "bases_per_s_per_gpu", "chained_pairs_per_s_per_gpu",
                                               "genomes_per_s_per_gpu", "phase_ms_per_step")}
            out["strong"].update({"collection": args.strong_collection, "genomes_per_gpu": sn, "chained_pairs": s["config"]["chained_pairs"], "order": "shuffled",
                                  "note": "the same N on a FIXED collection (BASELINE config 4's %d shuffled genomes, %d per GPU): speed-up over N = this block's ms_per_step at "
                                          "N = 1 / at N; every step's wall time on rank 0 is listed (step_wall_ms_rank0)" % (args.strong_collection, sn)})
            if "per_rank" in s:
                out["strong"]["per_rank"] = s["per_rank"]
    if world == 1 and default_shape and order == "clade" and not strong and not args.root_fasta and not args.no_variants and n_local == 1000 and args.strong_collection > 0:   # (--strong-collection 0 asks for the headline alone)
        # SURVEY 8d's secondary scale points on this code, this box: the dense collection (one clade: all 499,500 pairs chained), the other presets (cli.rs:60-68), 5,000 genomes
        variants = []
        for name, kw in (("dense: 1,000 genomes in ONE clade, every pair chained", dict(clade=1000, steps=2)), ("-c 30 (--slow)", dict(c=30, steps=3)), ("-c 70 (--medium)", dict(c=70, steps=5)),
                         ("-c 200 (--fast)", dict(c=200, steps=5)), ("5,000 genomes (250 clades of 20)", dict(n=5000, steps=3))):
            c0, cl0 = C, CLADE
            try:
                C, CLADE = kw.get("c", 125), kw.get("clade", 20)
                v = run_triangle(args, torch, dist, sk, ctx, None, None, 0, 1, device, kw.get("n", 1000), "clade", False, kw["steps"], 1, want_cpu=False, want_e2e=False)
                variants.append({"variant": name, "ms_per_step": v["ms_per_step"], "value": v["value"], "unit": v["unit"], "steps": v["steps"], "phase_ms_per_step": v["phase_ms_per_step"],
                                 "chained_pairs": v["config"]["chained_pairs"], "chained_pairs_per_s": v["chained_pairs_per_s_per_gpu"], "genomes": v["config"]["genomes"],
                                 "units": v.get("units"), "roofline_frac_seeding_hip_events": v["roofline"]["frac_hip_events"]})
            except Exception as e:
                variants.append({"variant": name, "error": repr(e)})
            finally:
                C, CLADE = c0, cl0
            torch.cuda.empty_cache()
        out["variants"] = variants
    if rank == 0:
        out.pop("_cpu_legs")()                                         # (cpu_baseline, e2e: behind every GPU measurement, see run_triangle)
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
