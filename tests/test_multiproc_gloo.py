"""N>1 path on CPU: two and four processes, gloo backend, the sharded triangle of skani_amd.distributed.  Compute goes through
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


def _test_genomes(interleave=True):
    """interleave: 3 clades x 4 members dealt out so that most candidate pairs cross the two rank blocks (sketches travel);
    otherwise 2 clades x 6, one per rank: no pair crosses and the exchange of sketches is skipped altogether."""
    from tests.parity_cases import synthetic_clades
    if not interleave:
        return synthetic_clades(n_clades=2, members=6, length=60000, seed=61, tiny=False)
    g = synthetic_clades(n_clades=3, members=4, length=60000, seed=51, tiny=False)
    return [g[(k % 3) * 4 + k // 3] for k in range(12)]


def _worker(rank, world, port, q, interleave=True):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import skani_amd as sk
    from skani_amd.distributed import distributed_triangle
    from tests.emu_lib import emu_lib
    from tests.parity_cases import synthetic_clades
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = sk.Context(0, lib=emu_lib())
        genomes = _test_genomes(interleave)
        per = len(genomes) // world
        mine = genomes[rank * per:(rank + 1) * per]
        params = sk.SketchParams()
        gs = ctx.pack_genomes([[s for _, s in g] for g in mine], params.seeding_mode)
        ss_local = ctx.sketch_genomes(gs, params, genome_rank=list(range(rank * per, (rank + 1) * per)))
        i, j, res, n = distributed_triangle(ctx, ss_local, params, sk.MapParams(learned_ani=True, compute_ci=True), dist, rank, world)
        if rank == 0:
            q.put((i, j, res, n))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("interleave,world", [(True, 2), (False, 2), (True, 4)])
def test_multi_rank_triangle_matches_single_process(interleave, world):
    import multiprocessing as mp
    import skani_amd as sk
    from tests.emu_lib import emu_lib
    from tests.helpers import MODEL_C125, ora
    from tests.parity_cases import assert_result_close, synthetic_clades
    emu_lib()   # build once before forking workers
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue(); port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q, interleave)) for r in range(world)]
    for p in procs:
        p.start()
    i, j, res, n = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    genomes = _test_genomes(interleave)
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
        genomes = _test_genomes(interleave)
        per = len(genomes) // world
        mine = genomes[rank * per:(rank + 1) * per]
        params = sk.SketchParams()
        gs = ctx.pack_genomes([[s for _, s in g] for g in mine], params.seeding_mode)
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
    genomes = _test_genomes(interleave)
    ctx = sk.Context(0)
    ss = ctx.sketch_records(genomes, sk.SketchParams(), None)
    si, sj, sres, sn = ctx.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
    assert sn == n and np.array_equal(si, i) and np.array_equal(sj, j) and sres.tobytes() == res.tobytes() and len(i) > 0
