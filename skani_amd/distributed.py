"""Multi-GPU triangle: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The path shards: sketching is per genome, chaining per pair.  Genomes are block-distributed; every rank
sketches its own block.  Exchange steps:
  1. all-gather of the MARKER sets + per-genome metadata only (~40 KB per 5 Mbp genome); every rank then screens ITS OWN
     rows against all genomes (an n_local x N count matrix: the per-rank screening cost stays flat as ranks are added);
  2. pair (i, j), i < j, is owned by the rank that owns genome i; a rank therefore needs the full sketch of a remote
     genome j only when a candidate pair crosses blocks.  The ranks tell each other which genomes they need (one small
     all-to-all) and exactly those sketches travel point-to-point (all-to-all of variable-size buffers).  For
     clade-structured collections almost nothing moves; in the worst case (every pair crosses) it degenerates to an
     all-gather of the raw sketches.
No collective inside the pair pipeline; the (small) results are gathered on rank 0."""
import pickle

import numpy as np

from . import _binding as B


def _all_to_all_bytes(dist, torch, device, payloads):
    """payloads[r] = bytes for rank r -> list of bytes received from every rank."""
    world = len(payloads)
    sizes = torch.tensor([len(p) for p in payloads], dtype=torch.int64, device=device)
    rsizes = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(rsizes, sizes)
    rs = [int(x) for x in rsizes.cpu()]
    send = torch.frombuffer(bytearray(b"".join(payloads)) or bytearray(1), dtype=torch.uint8)[:sum(len(p) for p in payloads)].to(device)
    recv = torch.empty(sum(rs), dtype=torch.uint8, device=device)
    dist.all_to_all_single(recv, send, output_split_sizes=rs, input_split_sizes=[len(p) for p in payloads])
    buf = recv.cpu().numpy().tobytes()
    out, o = [], 0
    for n in rs:
        out.append(buf[o:o + n]); o += n
    return out


def _gather_markers(ctx, ss_local, params, dist, rank, world, torch, device):
    """All-gather of the marker sets (flat arrays, no per-genome work): returns a markers-only SketchSet of ALL genomes whose
    genome order is rank order.  On GPUs the markers stay in device memory end to end (export -> RCCL all_gather -> import)."""
    n_local = len(ss_local)
    meta = ss_local.export_meta()
    _, M, NC = ss_local.totals()
    on_dev = device.type == "cuda"
    sizes = torch.tensor([M, NC], dtype=torch.int64, device=device)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    Ms = [int(x[0]) for x in all_sizes]
    max_m = max(max(Ms), 1)
    mk = torch.zeros(max_m, dtype=torch.int64, device=device)                       # u64 bit patterns
    if M:
        if on_dev:
            torch.cuda.synchronize(device)                                          # the library copies on its own stream: torch's fill must have landed
            ss_local.export_arrays(markers=mk.data_ptr(), device=True)
        else:
            ss_local.export_arrays(markers=mk.numpy().view(np.uint64))
    parts = [torch.empty_like(mk) for _ in range(world)]
    dist.all_gather(parts, mk)
    allmk = torch.cat([parts[r][:Ms[r]] for r in range(world)]) if sum(Ms) else torch.zeros(1, dtype=torch.int64, device=device)
    small = [None] * world                                                          # a few KB per rank: offsets, contig lengths, total lengths
    dist.all_gather_object(small, (meta["marker_off"], meta["contig_off"], meta["contig_lengths"], meta["total_len"]))
    n_total = n_local * world
    mo = np.zeros(n_total + 1, np.uint64); co = np.zeros(n_total + 1, np.uint64)
    g = 0
    for (m_off, c_off, _, _) in small:
        k = len(m_off) - 1
        mo[g + 1:g + k + 1] = mo[g] + m_off[1:]; co[g + 1:g + k + 1] = co[g] + c_off[1:]; g += k
    gmeta = dict(pos_off=np.zeros(n_total + 1, np.uint64), marker_off=mo, contig_off=co,
                 contig_lengths=np.concatenate([x[2] for x in small]).astype(np.uint32), total_len=np.concatenate([x[3] for x in small]).astype(np.uint64),
                 genome_rank=np.arange(n_total, dtype=np.uint32))
    if on_dev:
        torch.cuda.synchronize(device)
        return ctx.import_flat(params, gmeta, markers=allmk.data_ptr(), device=True), allmk
    return ctx.import_flat(params, gmeta, markers=allmk.numpy().view(np.uint64)), allmk


def distributed_triangle(ctx, ss_local, params, map_params, dist, rank, world, identity=0.0, rescue_small=True, torch=None, device=None):
    """ss_local: this rank's block of sketches, created with genome_rank = GLOBAL genome index; all blocks have the same
    size.  Returns (i, j, results, n_chained_total) with global indices, sorted by (i, j), on rank 0; (None, None, None, n)
    on the other ranks."""
    if world == 1:
        return ctx.triangle(ss_local, map_params, identity, rescue_small)
    if torch is None:
        import torch as _t
        torch = _t
    if device is None:
        device = torch.device("cpu")
    n_local = len(ss_local)
    base = rank * n_local
    # 1. markers + metadata of every genome, everywhere; this rank screens its rows (local genomes i) against all later
    #    genomes j > i: an n_local x N block of the triangle's screen (triangle.rs:55-90)
    markers_only, _keep = _gather_markers(ctx, ss_local, params, dist, rank, world, torch, device)
    gi, gj = ctx.screen_rows(markers_only, base, n_local, identity, rescue_small)
    markers_only.close()
    owner_j = gj // n_local
    # 2. sketches of remote partners j of my rows i: requests out, sketches back
    requests = [pickle.dumps(np.unique(gj[owner_j == r]) if r != rank else np.zeros(0, np.uint32), protocol=4) for r in range(world)]
    wanted = [pickle.loads(b) for b in _all_to_all_bytes(dist, torch, device, requests)]      # wanted[r]: my genomes that rank r needs
    payloads = [pickle.dumps([(int(g), ss_local.export(int(g) - base)) for g in lst], protocol=4) for lst in wanted]
    received = _all_to_all_bytes(dist, torch, device, payloads)
    remote = {}
    for blob in received:
        for g, rec in pickle.loads(blob):
            remote[g] = rec
    rem_ids = sorted(remote)
    rem_index = {g: k for k, g in enumerate(rem_ids)}
    li, lj = gi, gj
    local_pair = (lj // n_local) == rank
    res_parts = []
    n_chained = int(len(gi))
    if local_pair.any():
        r = ctx.chain_pairs(ss_local, None, li[local_pair] - base, lj[local_pair] - base, map_params)
        res_parts.append((li[local_pair], lj[local_pair], r))
    if (~local_pair).any():
        ss_rem = ctx.import_sketches(params, [remote[g] for g in rem_ids], genome_rank=np.array(rem_ids, dtype=np.uint32))
        qi = np.array([rem_index[int(g)] for g in lj[~local_pair]], dtype=np.uint32)
        r = ctx.chain_pairs(ss_local, ss_rem, li[~local_pair] - base, qi, map_params)      # ref = genome i (local), query = genome j (remote)
        res_parts.append((li[~local_pair], lj[~local_pair], r))
        ss_rem.close()
    if res_parts:
        ai = np.concatenate([p[0] for p in res_parts]); aj = np.concatenate([p[1] for p in res_parts]); ar = np.concatenate([p[2] for p in res_parts])
        keep = ar["ani"] > 0.1                                                       # triangle.rs:99
        ai, aj, ar = ai[keep], aj[keep], ar[keep]
    else:
        ai = np.zeros(0, np.uint32); aj = np.zeros(0, np.uint32); ar = np.zeros(0, B.RESULT_DTYPE)
    parts = [None] * world if rank == 0 else None
    dist.gather_object((ai, aj, ar, n_chained), parts, dst=0)
    if rank != 0:
        return None, None, None, n_chained
    ai = np.concatenate([p[0] for p in parts]); aj = np.concatenate([p[1] for p in parts])
    ar = np.concatenate([p[2] for p in parts]).view(B.RESULT_DTYPE)
    order = np.lexsort((aj, ai))
    return ai[order].astype(np.uint32), aj[order].astype(np.uint32), ar[order], sum(p[3] for p in parts)
