"""N>1 path: two to eight processes over gloo through skh_triangle_distributed (csrc/dist.hip).  On the CPU the compute goes through the
kernel simulator build (no GPU in this container) and the point is the protocol: screen rows, assignment, sketch exchange, result gather,
and that a failure on one rank stops all of them; the union of the ranks' work must equal the single-process triangle and the oracle.
The `-m gpu` tests at the end run the same protocol on an MI355X: BASELINE config 4's 10,000-genome collection as eight processes on one
device, and the library's own RCCL transport with a world of one."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


# case -> (genomes, number of genomes held by each rank)
def _case(case):
    if case.endswith("_root"): case = case[:-5]                                  # the same collection, result rows gathered on rank 0 only (SKH_DIST_ROWS_TO_ROOT)
    return _case_genomes(case)


def _case_genomes(case):
    """interleave / interleave4: 3 clades x 4 members dealt out so that most candidate pairs cross the rank blocks (sketches travel);
    blocks: 2 clades x 6, one per rank -- nothing has to move;
    uneven: 3 clades x 4 spread over four ranks holding 5, 0, 4 and 3 genomes (clades span the rank boundaries; one rank holds nothing);
    dense: ONE clade of 12 genomes on three ranks -- all 66 pairs are chained and have to be split evenly although they form one cluster;
    wide_one_rank: "interleave" with SKH_TUNE_WIDE_SPAN=0 on rank 0 only -- its sketch set is a wide one (64-bit coordinates, the form of genomes beyond
        2^31 padded bases), the other rank's is not: the ranks then exchange positions in the C ABI's form and every rank makes the set it chains its own way;
    wide_mixed: every other genome carries an extra contig that takes it past SKH_TUNE_WIDE_SPAN=90000 on all ranks: wide and ordinary genomes in every set."""
    from tests.parity_cases import synthetic_clades
    if case == "blocks":
        return synthetic_clades(n_clades=2, members=6, length=60000, seed=61, tiny=False), [6, 6]
    if case == "dense":
        from tests.helpers import mutate, random_genome
        root = random_genome(40000, 71)
        return [[("c0", mutate(root, 0.004 + 0.003 * m, 7100 + m))] for m in range(12)], [4, 4, 4]
    if case == "interleave8":          # eight ranks (the node size of BASELINE config 4) x two genomes; every clade is spread over four ranks
        g = synthetic_clades(n_clades=4, members=4, length=40000, seed=81, tiny=False)
        return [g[(k % 4) * 4 + k // 4] for k in range(16)], [2] * 8
    g = synthetic_clades(n_clades=3, members=4, length=60000, seed=51, tiny=False)
    g = [g[(k % 3) * 4 + k // 3] for k in range(12)]
    if case == "wide_mixed":
        from tests.helpers import random_genome
        g = [x + [("extra%d" % k, random_genome(30000, 900 + k))] if k % 2 == 0 else x for k, x in enumerate(g)]
    return g, {"interleave": [6, 6], "interleave4": [3, 3, 3, 3], "uneven": [5, 0, 4, 3], "wide_one_rank": [6, 6], "wide_mixed": [4, 4, 4]}[case]


def _wide_span_of(case, rank):
    return {"wide_one_rank": "0" if rank == 0 else None, "wide_mixed": "90000"}.get(case)


def _worker(rank, world, port, q, case):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import skani_amd as sk
    from skani_amd.distributed import distributed_triangle
    from tests.emu_lib import emu_lib
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if _wide_span_of(case, rank) is not None: os.environ["SKH_TUNE_WIDE_SPAN"] = _wide_span_of(case, rank)
        to_root = case.endswith("_root"); case = case[:-5] if to_root else case
        if case == "interleave8": os.environ["SKH_TUNE_SCREEN_CELLS_DENSE"] = "1"   # the gathered cells through the dense count matrix (every other case: row by row in LDS)
        if case == "uneven": os.environ["SKH_TUNE_SCREEN_CELLS"] = "40"      # a count matrix of 12 x 12 cells does not fit: the screen is cut by rows (the form of very large collections)
        ctx = sk.Context(0, lib=emu_lib())
        genomes, held = _case(case)
        base = sum(held[:rank])
        mine = genomes[base:base + held[rank]]
        params = sk.SketchParams()
        gs = ctx.pack_genomes([[s for _, s in g if len(s) >= 500] for g in mine], params.seeding_mode)      # file_io.rs:176
        # half of the cases sketch with deferred seed tables (what bench.py does on several GPUs): a rank then indexes only the sketches it chains
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=list(range(base, base + held[rank])), defer_tables=case in ("interleave4", "uneven", "dense", "blocks", "interleave8", "wide_mixed"))
        if case.startswith("wide"): assert ss_local.wide == (case == "wide_mixed" or rank == 0)
        i, j, res, n, st = distributed_triangle(ctx, ss_local, params, sk.MapParams(learned_ani=True, compute_ci=True), dist, rank, world, with_stats=True, rows_to_root=to_root)
        q.put((rank, i, j, res, n, st))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["interleave", "blocks", "interleave4", "uneven", "dense", "interleave8", "wide_one_rank", "wide_mixed", "interleave4_root", "uneven_root", "dense_root"])
def test_multi_rank_triangle_matches_single_process(case):
    import multiprocessing as mp
    import skani_amd as sk
    from tests.emu_lib import emu_lib
    from tests.helpers import MODEL_C125, ora
    from tests.parity_cases import assert_result_close
    emu_lib()   # build once before forking workers
    genomes, held = _case(case); world = len(held)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue(); port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    _, i, j, res, n, st0 = got[0]
    to_root = case.endswith("_root"); case = case[:-5] if to_root else case
    if to_root:                    # rank 0 returns the whole triangle, every other rank the rows of the pairs it chained: disjoint stretches of it, in (i, j) order
        where = {(int(a), int(b)): x for x, (a, b) in enumerate(zip(i, j))}; seen = set(); others = 0
        for r in range(1, world):
            ri, rj, rres = got[r][1], got[r][2], got[r][3]
            assert got[r][4] == n and len(ri) <= got[r][5]["n_pairs_mine"]
            keys = [(int(a), int(b)) for a, b in zip(ri, rj)]
            assert keys == sorted(keys) and not (set(keys) & seen) and all(k in where for k in keys)
            assert all(rres[x].tobytes() == res[where[k]].tobytes() for x, k in enumerate(keys))
            seen |= set(keys); others += len(keys)
        assert others < len(i) or got[0][5]["n_pairs_mine"] == 0
    for r in range(1, world if not to_root else 1):      # every rank returns the whole triangle
        assert np.array_equal(got[r][1], i) and np.array_equal(got[r][2], j) and got[r][3].tobytes() == res.tobytes() and got[r][4] == n
    # oracle: genome ranks are the global indices (names sort like indices)
    osk = [ora.sketch_records(g, file_name="g%03d" % k) for k, g in enumerate(genomes)]
    oi, oj, ores, onch, _ = ora.triangle(osk, model=ora.Model(MODEL_C125))
    assert n == onch and np.array_equal(i, oi) and np.array_equal(j, oj)
    for x in range(len(res)):
        assert_result_close(res[x], ores[x], (int(i[x]), int(j[x])))
    # single-process result through the same library
    ctx = sk.Context(0, lib=emu_lib())
    ss = ctx.sketch_records(genomes, sk.SketchParams(), None)
    si, sj, sres, sn = ctx.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
    assert np.array_equal(si, i) and np.array_equal(sj, j) and sres.tobytes() == res.tobytes()
    # the shares: every candidate pair is chained exactly once, the screen rows tile [0, N)
    stats = [g[5] for g in got]
    assert sum(s["n_pairs_mine"] for s in stats) == n and all(s["n_candidate_pairs_total"] == n and s["n_genomes_total"] == len(genomes) for s in stats)
    assert all(s["screen_by_key_range"] == (0 if case == "uneven" else 1) for s in stats)
    # the marker sets: cut by key range, a rank receives its own part of the range from the others -- the parts together are every marker once; the row form gathers all
    all_markers = 8 * sum(o.n_markers for o in osk)
    got_markers = sum(s["marker_bytes_received"] for s in stats)
    if case == "uneven":
        assert got_markers == (world - 1) * all_markers
    else:
        assert got_markers < all_markers and got_markers >= all_markers * (world - 1) // world - 8 * len(osk) * world
    assert stats[0]["screen_row_begin"] == 0 and stats[-1]["screen_row_end"] == len(genomes)
    assert all(stats[r]["screen_row_end"] == stats[r + 1]["screen_row_begin"] for r in range(world - 1))
    if case == "blocks":        # one cluster per rank: nothing travels
        assert all(s["bytes_received"] == 0 and s["n_genomes_received"] == 0 for s in stats)
    if case in ("interleave", "interleave4", "uneven", "interleave8", "wide_one_rank", "wide_mixed"):
        assert sum(s["n_genomes_received"] for s in stats) > 0 and sum(s["bytes_sent"] for s in stats) == sum(s["bytes_received"] for s in stats) > 0
    if case == "dense":         # one cluster of 66 pairs over three ranks: cut into tiles, shares within 10 % of the mean
        mean = n / world
        assert n == 66 and all(abs(s["n_pairs_mine"] - mean) <= 0.1 * mean for s in stats), [s["n_pairs_mine"] for s in stats]


def _failing_worker(rank, world, port, q, fail_rank, phase):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if rank == fail_rank:
        os.environ["SKH_TUNE_DIST_FAIL"] = str(phase)
    import torch.distributed as dist
    import skani_amd as sk
    from skani_amd.distributed import distributed_triangle
    from tests.emu_lib import emu_lib
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = sk.Context(0, lib=emu_lib())
        genomes, held = _case("interleave4")
        base = sum(held[:rank])
        params = sk.SketchParams()
        gs = ctx.pack_genomes([[s for _, s in g if len(s) >= 500] for g in genomes[base:base + held[rank]]], params.seeding_mode)
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=list(range(base, base + held[rank])), defer_tables=True)
        try:
            distributed_triangle(ctx, ss_local, params, sk.MapParams(), dist, rank, world)
            q.put((rank, "no error"))
        except sk.SkaniHipError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("phase", [1, 3, 4, 5, 7, 8, 9, 10])
def test_failure_on_one_rank_stops_every_rank(phase):
    """One rank fails in a local phase (injected: marker buffers, the screen of its key range, the candidate list from the gathered cells, exchange buffers,
    home tables, home pairs -- while the sketch exchange is in flight --, received sketches, away pairs): at the next exchange point all ranks agree on it
    and return an error -- nobody waits in a collective for a rank that has left, and the asynchronous exchange is closed on every path (dist.hip `agree`,
    ExchangeGuard)."""
    import multiprocessing as mp
    from tests.emu_lib import emu_lib
    emu_lib()
    world, fail_rank = 4, 2
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue(); port = _free_port()
    procs = [ctxm.Process(target=_failing_worker, args=(r, world, port, q, fail_rank, phase)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    assert "injected failure" in got[fail_rank] and "skani_hip error -5" not in got[fail_rank], got   # the rank that failed keeps its own code
    for r in range(world):
        if r != fail_rank:
            assert "rank %d failed" % fail_rank in got[r] and "skani_hip error -5:" in got[r], got        # SKH_ERR_PEER: another rank's failure


def _reuse_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import skani_amd as sk
    from skani_amd.distributed import Comm
    from tests.emu_lib import emu_lib
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = sk.Context(0, lib=emu_lib())
        genomes, held = _case("interleave4")
        base = sum(held[:rank])
        params = sk.SketchParams()
        comm = Comm.host(ctx, dist, rank, world)
        comm.selftest()
        out = []
        # call 1: every rank holds ONE genome (small tables); calls 2 and 3: the whole collection -- the communicator's remembered capacities are too small
        # in call 2 (every gather takes its second round) and exact in call 3 (one round each)
        for n_mine in (1, held[rank], held[rank]):
            mine = genomes[base:base + n_mine]
            gs = ctx.pack_genomes([[s for _, s in g if len(s) >= 500] for g in mine], params.seeding_mode)
            ss = ctx.sketch_genomes(gs, params, genome_rank=list(range(base, base + n_mine)), defer_tables=True)
            i, j, res, n, st = comm.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
            out.append((i, j, res, n, st))
            ss.close(); gs.close()
        comm.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_communicator_reused_with_growing_and_equal_sizes():
    """The gathers of skh_triangle_distributed are laid out by the sizes of the communicator's previous call (one collective instead of counts + payload):
    a first call with one genome per rank, a second with the whole collection (every gather has to go round twice), a third with the same sizes (one
    round).  Calls 2 and 3 return the single-process triangle byte by byte; call 1 the triangle of its four genomes.  The communicator's self-test
    (host and device all-gather, blocking and asynchronous all-to-all) runs first."""
    import multiprocessing as mp
    import skani_amd as sk
    from tests.emu_lib import emu_lib
    emu_lib()
    genomes, held = _case("interleave4"); world = len(held)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue(); port = _free_port()
    procs = [ctxm.Process(target=_reuse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    ctx = sk.Context(0, lib=emu_lib())
    mp_ = sk.MapParams(learned_ani=True, compute_ci=True)
    firsts = [genomes[sum(held[:r])] for r in range(world)]
    for call, gset in ((0, firsts), (1, genomes), (2, genomes)):
        ss = ctx.sketch_records(gset, sk.SketchParams(), None)
        si, sj, sres, sn = ctx.triangle(ss, mp_)
        for r in range(world):
            i, j, res, n, st = got[r][call]
            assert n == sn and np.array_equal(i, si) and np.array_equal(j, sj) and res.tobytes() == sres.tobytes(), (call, r)
        ss.close()
    assert sum(got[r][2][4]["n_pairs_home"] for r in range(world)) < got[0][2][3]          # interleaved clades: most pairs need a received sketch
    ctx.close()


def test_plan_balances_config4_shaped_collection():
    """BASELINE config 4's shape without the genomes: 10,000 genomes in clades of 20, file order shuffled (file_io.rs:250 sorts by NAME, not by
    clade), eight ranks of 1,250.  The plan keeps clusters whole, gives every rank the same number of pairs within 2 %, and moves every sketch
    at most once; the old row-ownership rule gave rank 0 about 23 % of the pairs of a dense collection."""
    from skani_amd.distributed import plan_pairs
    from tests.emu_lib import emu_lib
    L = emu_lib()
    rng = np.random.default_rng(4)
    N, world, per = 10000, 8, 1250
    clade = rng.permutation(N) // 20                       # clade of global genome g
    order = np.argsort(clade, kind="stable")
    pi, pj = [], []
    for c0 in range(0, N, 20):
        m = np.sort(order[c0:c0 + 20])
        a, b = np.triu_indices(20, 1)
        pi.append(m[a]); pj.append(m[b])
    pi = np.concatenate(pi).astype(np.uint32); pj = np.concatenate(pj).astype(np.uint32)
    o = np.lexsort((pj, pi)); pi, pj = pi[o], pj[o]
    w = rng.integers(4500, 5500, N).astype(np.uint64)
    holder = np.arange(N) // per
    owner = plan_pairs(L, N, pi, pj, w, world, holder)
    counts = np.bincount(owner, minlength=world)
    assert counts.sum() == 95000 and counts.max() - counts.min() <= 0.02 * counts.mean(), counts
    assert all(len(set(owner[(clade[pi] == c)])) == 1 for c in range(0, 500, 37))          # clusters stay whole
    need = np.zeros((N, world), bool); need[pi, owner] = True; need[pj, owner] = True
    moved = need.sum() - need[np.arange(N), holder].sum()
    assert moved <= N and need.sum(1).max() == 1                                             # every sketch is needed by exactly one rank
    # a dense collection (every pair a candidate): tiles, shares within 10 %
    n2 = 400
    a, b = np.triu_indices(n2, 1)
    dense_owner = plan_pairs(L, n2, a.astype(np.uint32), b.astype(np.uint32), np.full(n2, 5000, np.uint64), world)
    counts = np.bincount(dense_owner, minlength=world)
    assert counts.max() - counts.min() <= 0.1 * counts.mean(), counts
    # a collection whose clusters sit on one rank each (sorted by clade): the plan leaves every cluster at home, nothing travels
    cl2 = np.arange(N) // 20
    pi2 = np.concatenate([c0 + np.triu_indices(20, 1)[0] for c0 in range(0, N, 20)]).astype(np.uint32)
    pj2 = np.concatenate([c0 + np.triu_indices(20, 1)[1] for c0 in range(0, N, 20)]).astype(np.uint32)
    hold2 = (np.arange(N) // 1260).astype(np.uint32)             # 63 whole clades per rank (the last rank gets fewer)
    owner = plan_pairs(L, N, pi2, pj2, w, world, hold2)
    assert np.mean(owner == hold2[pi2]) > 0.9 and np.bincount(owner, minlength=world).max() <= 1.05 * 95000 / world
    # deterministic: the same inputs give the same plan (every rank computes it for itself)
    assert np.array_equal(dense_owner, plan_pairs(L, n2, a.astype(np.uint32), b.astype(np.uint32), np.full(n2, 5000, np.uint64), world))


def _two_rank_genomes(interleave):
    """Two clades of six 1.2 Mbp genomes (1-6 contigs each).  interleave: the clades alternate, so every clade has members on both ranks and whole
    sketches cross; otherwise one clade per rank and nothing moves."""
    from tests.parity_cases import synthetic_clades
    g = synthetic_clades(n_clades=2, members=6, length=1_200_000, seed=61, tiny=False)
    return [g[(k % 2) * 6 + k // 2] for k in range(12)] if interleave else g


def _gpu_worker(rank, world, port, q, interleave=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import skani_amd as sk
    from skani_amd.distributed import distributed_triangle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = sk.Context(0)
        genomes = _two_rank_genomes(interleave)
        per = len(genomes) // world
        mine = genomes[rank * per:(rank + 1) * per]
        params = sk.SketchParams()
        gs = ctx.pack_genomes([[s for _, s in g if len(s) >= 500] for g in mine], params.seeding_mode)
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=list(range(rank * per, (rank + 1) * per)), defer_tables=interleave)
        i, j, res, n, st = distributed_triangle(ctx, ss_local, params, sk.MapParams(learned_ani=True, compute_ci=True), dist, rank, world, torch=torch, with_stats=True)
        q.put((rank, i, j, res, n, st))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("interleave", [False, True])
def test_two_ranks_on_one_gpu(interleave):
    """skh_triangle_distributed with two processes sharing the one GPU (host-collective transport over gloo; RCCL needs a GPU per rank): 12 genomes
    of 1.2 Mbp.  interleave: every clade has members on both ranks, so whole sketches travel through the all-to-all and are indexed where they
    arrive.  Checked against the oracle field by field and against the single-process triangle byte by byte."""
    import multiprocessing as mp
    import skani_amd as sk
    from tests.helpers import MODEL_C125, ora
    from tests.parity_cases import assert_result_close
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue(); port = _free_port()
    procs = [ctxm.Process(target=_gpu_worker, args=(r, 2, port, q, interleave)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    _, i, j, res, n, st0 = got[0]
    assert np.array_equal(got[1][1], i) and np.array_equal(got[1][2], j) and got[1][3].tobytes() == res.tobytes() and got[1][4] == n
    genomes = _two_rank_genomes(interleave)
    osk = [ora.sketch_records(g, file_name="g%03d" % k) for k, g in enumerate(genomes)]
    oi, oj, ores, onch, _ = ora.triangle(osk, model=ora.Model(MODEL_C125))
    assert n == onch == 30 and np.array_equal(i, oi) and np.array_equal(j, oj)
    for x in range(len(res)):
        assert_result_close(res[x], ores[x], (int(i[x]), int(j[x])))
    ctx = sk.Context(0)
    ss = ctx.sketch_records(genomes, sk.SketchParams(), None)
    si, sj, sres, sn = ctx.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
    assert sn == n and np.array_equal(si, i) and np.array_equal(sj, j) and sres.tobytes() == res.tobytes() and len(i) > 0
    moved = sum(g[5]["n_genomes_received"] for g in got)
    assert (moved > 0 and sum(g[5]["bytes_sent"] for g in got) == sum(g[5]["bytes_received"] for g in got) > 0) if interleave else moved == 0
    ss.close(); ctx.close()


def _config4_worker(rank, world, port, q, n_local):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import hashlib
    import torch
    import torch.distributed as dist
    import bench
    import skani_amd as sk
    from skani_amd.distributed import Comm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        ctx = sk.Context(0)
        canon = bench.genome_order(n_local * world, "shuffled")
        bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, canon[rank * n_local:(rank + 1) * n_local])
        torch.cuda.synchronize()
        gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
        del bases
        torch.cuda.empty_cache()
        ss = ctx.sketch_genomes(gs, sk.SketchParams(), genome_rank=np.arange(rank * n_local, (rank + 1) * n_local, dtype=np.uint32), defer_tables=True)
        gs.close()
        comm = Comm.host(ctx, dist, rank, world, torch=torch)
        i, j, res, n, st = comm.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
        digest = hashlib.sha256(i.tobytes() + j.tobytes() + res.tobytes()).hexdigest()
        q.put((rank, digest, n, st, (i, j, res) if rank == 0 else None))
        comm.close(); ss.close(); ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_config4_eight_ranks_one_device():
    """BASELINE config 4 -- the triangle over 10,000 synthetic ~5 Mbp genomes tiled over eight ranks -- with the eight ranks as eight processes on
    the ONE GPU of this box (host-collective transport; bench.py --gpus 8 runs the same call over RCCL, one GPU per rank).  Every rank generates and
    sketches its 1,250 genomes of bench.py's shuffled collection (deferred seed tables) and calls skh_triangle_distributed.  Checked: 95,000
    chained pairs, all inside clades; the same result bytes on every rank; cost shares within 1 % (pair counts within 3 %); what is sent is received, no sketch travels twice;
    one clade field by field against the oracle; and the whole result byte by byte against ONE process running skh_triangle over the same 10,000
    genomes on the same GPU (50 GB of ASCII, 12.5 GB packed, ~14 GB of sketches)."""
    import multiprocessing as mp
    import hashlib
    import torch
    import bench
    import skani_amd as sk
    from tests.helpers import MODEL_C125, ora
    from tests.parity_cases import assert_result_close
    world, n_local = 8, 1250
    N = world * n_local
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue(); port = _free_port()
    procs = [ctxm.Process(target=_config4_worker, args=(r, world, port, q, n_local)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=1500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300); assert p.exitcode == 0
    i, j, res = got[0][4]
    canon = bench.genome_order(N, "shuffled")
    clade_of = canon // 20                                         # clade of global genome g
    assert all(g[2] == 95000 for g in got) and len(i) == 95000 and (i < j).all() and (clade_of[i] == clade_of[j]).all()
    assert len({g[1] for g in got}) == 1                           # identical result bytes on every rank
    stats = [g[3] for g in got]
    shares = np.array([s["n_pairs_mine"] for s in stats]); costs = np.array([s["cost_mine"] for s in stats], np.float64)
    # whole clusters (190 pairs each) are dealt out by estimated cost (marker counts: the genomes are 4.5-5.5 Mbp): costs within 1 %, pair counts within 3 %
    assert shares.sum() == 95000 and np.abs(shares - shares.mean()).max() <= 0.03 * shares.mean(), shares
    assert np.abs(costs - costs.mean()).max() <= 0.01 * costs.mean() and all(s["cost_total"] == costs.sum() for s in stats), costs
    sent, recvd, moved = sum(s["bytes_sent"] for s in stats), sum(s["bytes_received"] for s in stats), sum(s["n_genomes_received"] for s in stats)
    assert sent == recvd > 0 and 0 < moved <= N                    # every sketch is needed by one rank only: it travels at most once
    assert stats[0]["screen_row_begin"] == 0 and stats[-1]["screen_row_end"] == N and all(stats[r]["screen_row_end"] == stats[r + 1]["screen_row_begin"] for r in range(world - 1))
    # one clade against the oracle (host copies of the same bytes; names sort like the global indices)
    dev = torch.device("cuda", 0)
    members = np.nonzero(clade_of == 137)[0]
    _, _, _, _, host = bench.make_genomes(torch, dev, canon[members], keep_host=True)
    osk = [ora.sketch_records(g, file_name="s%05d.fa" % int(members[k])) for k, g in enumerate(host)]
    oi, oj, ores, onch, _ = ora.triangle(osk, model=ora.Model(MODEL_C125))
    sel = clade_of[i] == 137
    assert onch == 190 and np.array_equal(i[sel], members[oi]) and np.array_equal(j[sel], members[oj])
    for x, y in zip(res[sel], ores):
        assert_result_close(x, y)
    # the same collection in one process
    ctx = sk.Context(0)
    bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, canon)
    torch.cuda.synchronize()
    gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
    del bases
    torch.cuda.empty_cache()
    ss = ctx.sketch_genomes(gs, sk.SketchParams(), genome_rank=np.arange(N, dtype=np.uint32))
    gs.close()
    si, sj, sres, sn = ctx.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
    assert sn == 95000 and hashlib.sha256(si.tobytes() + sj.tobytes() + sres.tobytes()).hexdigest() == got[0][1]
    ss.close(); ctx.close()
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_rccl_world_size_one():
    """The library's own RCCL transport (csrc/rccl_transport.hip: librccl loaded with dlopen, communicator on the context's stream) with a world
    of one -- all a one-GPU box can offer: skh_comm_unique_id, skh_comm_create_rccl, every all-gather and the (empty) send/recv group of
    skh_triangle_distributed run through RCCL.  40 genomes of ~1 Mbp; the result equals skh_triangle byte by byte and the oracle field by field."""
    import torch
    import bench
    import skani_amd as sk
    from skani_amd.distributed import Comm
    from tests.helpers import MODEL_C125, ora
    from tests.parity_cases import assert_result_close
    dev = torch.device("cuda", 0)
    ctx = sk.Context(0)
    try:
        bases, coff, cgen, ng, host = bench.make_genomes(torch, dev, np.arange(40), mean_len=1_000_000, keep_host=True)
        torch.cuda.synchronize()
        gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
        del bases
        ss = ctx.sketch_genomes(gs, sk.SketchParams(), genome_rank=np.arange(ng, dtype=np.uint32), defer_tables=True)
        gs.close()
        mp_ = sk.MapParams(learned_ani=True, compute_ci=True)
        comm = Comm.rccl(ctx, None, 0, 1, torch=torch, device=dev)
        i, j, res, n, st = comm.triangle(ss, mp_)
        comm.close()
        assert "librccl" in open("/proc/self/maps").read()
        si, sj, sres, sn = ctx.triangle(ss, mp_)
        assert n == sn == 380 and np.array_equal(i, si) and np.array_equal(j, sj) and res.tobytes() == sres.tobytes()
        assert st["n_pairs_mine"] == 380 and st["n_genomes_received"] == 0 and st["screen_row_begin"] == 0 and st["screen_row_end"] == 40 and st["screen_by_key_range"] == 0
        osk = [ora.sketch_records(g, file_name="s%05d.fa" % k) for k, g in enumerate(host)]
        oi, oj, ores, onch, _ = ora.triangle(osk, model=ora.Model(MODEL_C125))
        assert onch == 380 and np.array_equal(i, oi) and np.array_equal(j, oj)
        for x in range(len(res)):
            assert_result_close(res[x], ores[x], (int(i[x]), int(j[x])))
        # the screen by key range through RCCL (a world of one takes the row form unless told otherwise): the cells are gathered in the communicator's own device
        # buffers -- counts first and buffers made in its first call, one device all-gather in the second
        os.environ["SKH_TUNE_DIST_KEY_RANGE_W1"] = "1"
        try:
            ctx2 = sk.Context(0)
        finally:
            del os.environ["SKH_TUNE_DIST_KEY_RANGE_W1"]
        try:
            meta = ss.export_meta(); n_pos, n_mk, _ = ss.totals()
            arrs = dict(seed=np.zeros(n_pos, np.uint32), pos=np.zeros(n_pos, np.uint32), ctgcanon=np.zeros(n_pos, np.uint32), markers=np.zeros(n_mk, np.uint64))
            ss.export_arrays(**arrs)
            ss2 = ctx2.import_flat(ss.params, meta, **arrs)
            comm2 = Comm.rccl(ctx2, None, 0, 1, torch=torch, device=dev)
            for call in range(2):
                i2, j2, res2, n2, st2 = comm2.triangle(ss2, mp_)
                assert st2["screen_by_key_range"] == 1 and n2 == 380 and np.array_equal(i2, si) and np.array_equal(j2, sj) and res2.tobytes() == sres.tobytes(), call
            comm2.close(); ss2.close()
        finally:
            ctx2.close()
        ss.close()
    finally:
        ctx.close()
        torch.cuda.empty_cache()
