"""N>1 path on CPU: two to eight processes, gloo backend, the sharded triangle of skani_amd.distributed.  Compute goes through
the kernel simulator build (no GPU in this container); the point of the test is the sharding / exchange / gather logic:
the union of the two ranks' work must equal the single-process triangle and the oracle."""
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
    """interleave / interleave4: 3 clades x 4 members dealt out so that most candidate pairs cross the rank blocks (sketches travel);
    blocks: 2 clades x 6, one per rank -- nothing has to move;
    uneven: 3 clades x 4 spread over four ranks holding 5, 0, 4 and 3 genomes (clades span the rank boundaries; one rank holds nothing);
    dense: ONE clade of 12 genomes on three ranks -- all 66 pairs are chained and have to be split evenly although they form one cluster."""
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
    return g, {"interleave": [6, 6], "interleave4": [3, 3, 3, 3], "uneven": [5, 0, 4, 3]}[case]


def _worker(rank, world, port, q, case):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import skani_amd as sk
    from skani_amd.distributed import distributed_triangle
    from tests.emu_lib import emu_lib
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = sk.Context(0, lib=emu_lib())
        genomes, held = _case(case)
        base = sum(held[:rank])
        mine = genomes[base:base + held[rank]]
        params = sk.SketchParams()
        gs = ctx.pack_genomes([[s for _, s in g if len(s) >= 500] for g in mine], params.seeding_mode)      # file_io.rs:176
        # half of the cases sketch with deferred seed tables (what bench.py does on several GPUs): a rank then indexes only the sketches it chains
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=list(range(base, base + held[rank])), defer_tables=case in ("interleave4", "uneven", "dense", "blocks", "interleave8"))
        i, j, res, n, st = distributed_triangle(ctx, ss_local, params, sk.MapParams(learned_ani=True, compute_ci=True), dist, rank, world, with_stats=True)
        q.put((rank, i, j, res, n, st))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["interleave", "blocks", "interleave4", "uneven", "dense", "interleave8"])
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
    for r in range(1, world):      # every rank returns the whole triangle
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
    assert stats[0]["screen_row_begin"] == 0 and stats[-1]["screen_row_end"] == len(genomes)
    assert all(stats[r]["screen_row_end"] == stats[r + 1]["screen_row_begin"] for r in range(world - 1))
    if case == "blocks":        # one cluster per rank: nothing travels
        assert all(s["bytes_received"] == 0 and s["n_genomes_received"] == 0 for s in stats)
    if case in ("interleave", "interleave4", "uneven", "interleave8"):
        assert sum(s["n_genomes_received"] for s in stats) > 0 and sum(s["bytes_sent"] for s in stats) == sum(s["bytes_received"] for s in stats) > 0
    if case == "dense":         # one cluster of 66 pairs over three ranks: cut into tiles, shares within 10 % of the mean
        mean = n / world
        assert n == 66 and all(abs(s["n_pairs_mine"] - mean) <= 0.1 * mean for s in stats), [s["n_pairs_mine"] for s in stats]


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


def _gpu_worker(rank, world, port, q, interleave=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import skani_amd as sk
    from skani_amd.distributed import distributed_triangle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        ctx = sk.Context(0)
        genomes, held = _case("interleave" if interleave else "blocks")
        per = held[0]
        mine = genomes[rank * per:(rank + 1) * per]
        params = sk.SketchParams()
        gs = ctx.pack_genomes([[s for _, s in g if len(s) >= 500] for g in mine], params.seeding_mode)
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=list(range(rank * per, (rank + 1) * per)))
        i, j, res, n = distributed_triangle(ctx, ss_local, params, sk.MapParams(learned_ani=True, compute_ci=True), dist, rank, world, torch=torch, device=dev)
        if rank == 0:
            q.put((i, j, res, n))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("interleave", [False, True])
def test_two_ranks_device_tensors_on_one_gpu(interleave):
    """The device-memory path of the exchange (export into torch CUDA tensors -> all_gather -> import from the gathered tensor,
    results gathered as CUDA byte tensors; interleave: pairs cross the rank blocks, so whole sketches travel as CUDA tensors through
    all_to_all_single): two processes share the one GPU, collectives by gloo (RCCL needs one GPU per rank)."""
    import multiprocessing as mp
    import skani_amd as sk
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue(); port = _free_port()
    procs = [ctxm.Process(target=_gpu_worker, args=(r, 2, port, q, interleave)) for r in range(2)]
    for p in procs:
        p.start()
    i, j, res, n = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    genomes, _ = _case("interleave" if interleave else "blocks")
    ctx = sk.Context(0)
    ss = ctx.sketch_records(genomes, sk.SketchParams(), None)
    si, sj, sres, sn = ctx.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
    assert sn == n and np.array_equal(si, i) and np.array_equal(sj, j) and sres.tobytes() == res.tobytes() and len(i) > 0
