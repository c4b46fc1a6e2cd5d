"""Multi-GPU triangle: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The path shards: sketching is per genome, chaining per pair.  The only exchange is an all-gather of the
raw sketches so that every rank holds the full set; after that the screened pair list is split round-robin and there is
no collective until the (small) result gather on rank 0."""
import numpy as np

from . import _binding as B


def exchange_sketches(ctx, ss_local, params, dist, world):
    """All-gather position-ordered seeds, markers and contig tables of every rank's genomes; rebuild the derived
    tables (seed order, CSR, hash tables) locally.  Genome order = rank order, so global ids are contiguous blocks."""
    if world == 1:
        return ss_local
    per = [ss_local.export(g) for g in range(len(ss_local))]
    gathered = [None] * world
    dist.all_gather_object(gathered, per)
    allg = [d for part in gathered for d in part]
    return ctx.import_sketches(params, allg, genome_rank=np.arange(len(allg), dtype=np.uint32))


def distributed_triangle(ctx, ss_all, map_params, dist, rank, world, identity=0.0, rescue_small=True):
    """Every rank screens the full set, chains pairs rank, rank+world, ... and rank 0 receives all kept results
    sorted by (i, j).  Returns (i, j, results, n_chained_total) on rank 0 and (None, None, None, n) elsewhere."""
    i, j, res, n_chained = ctx.triangle(ss_all, map_params, identity, rescue_small, part=rank, n_parts=world)
    if world == 1:
        return i, j, res, n_chained
    parts = [None] * world if rank == 0 else None
    dist.gather_object((i, j, res, n_chained), parts, dst=0)
    if rank != 0:
        return None, None, None, n_chained
    ai = np.concatenate([p[0] for p in parts]); aj = np.concatenate([p[1] for p in parts])
    ar = np.concatenate([p[2] for p in parts]).view(B.RESULT_DTYPE)
    order = np.lexsort((aj, ai))
    return ai[order], aj[order], ar[order], sum(p[3] for p in parts)
