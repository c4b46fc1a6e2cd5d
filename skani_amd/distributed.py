"""Multi-GPU triangle: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The path shards: sketching is per genome, chaining per pair.  Genomes are block-distributed; every rank
sketches its own block.  Exchange steps:
  1. all-gather of the MARKER sets + two numbers per genome (~40 KB per 5 Mbp genome); every rank then screens ITS OWN
     rows against all later genomes (an n_local x N block of the triangle's screen: the per-rank screening cost stays
     nearly flat as ranks are added);
  2. pair (i, j), i < j, is owned by the rank that owns genome i; a rank therefore needs the full sketch of a remote
     genome j only when a candidate pair crosses blocks.  One all-reduce tells whether any pair does; only then do the ranks
     tell each other which genomes they need and exactly those sketches travel point-to-point, as device tensors
     (all_to_all_single with variable splits).  For clade-structured collections nothing moves; in the worst case (every pair crosses) it
     degenerates to an all-gather of the raw sketches.
No collective inside the pair pipeline; the (small) results are all-gathered at the end.  Every collective is a tensor
collective (all_gather / all_reduce / all_to_all_single): five per triangle in the common case."""
import numpy as np

from . import _binding as B


def _all_gather_padded(dist, torch, device, t, sizes):
    """all-gather of 1-D tensors of different lengths (sizes[r] known everywhere): pad to the longest, gather, cut."""
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=device)
    if t.numel():
        pad[:t.numel()] = t
    parts = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(parts, pad)
    return [parts[r][:sizes[r]] for r in range(len(sizes))]


def _all_to_all_tensor(dist, torch, device, send, send_counts):
    """send: 1-D tensor laid out by destination rank, send_counts[r] elements for rank r -> (received tensor, counts per source)."""
    world = len(send_counts)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=device)
    rc = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(rc, sc)
    rcs = [int(x) for x in rc.cpu()]
    recv = torch.empty(sum(rcs), dtype=send.dtype, device=device)
    dist.all_to_all_single(recv, send, output_split_sizes=rcs, input_split_sizes=[int(x) for x in send_counts])
    return recv, rcs


def _exchange_sketches(ctx, ss_local, params, need, base, dist, rank, world, torch, device):
    """Point-to-point exchange of whole sketches: need[r] = sorted global ids of rank r's genomes this rank chains against.
    Returns (SketchSet of the received genomes in increasing global id, their ids).  The sketch arrays travel as device tensors
    (export_flat -> slices -> all_to_all_single over RCCL -> import_flat): three payload collectives -- the 32-bit arrays
    (seed | position | contig-and-strand per genome block), the 64-bit markers, and a small 64-bit header with the per-genome sizes."""
    on_dev = device.type == "cuda"
    # which of MY genomes every other rank wants (control plane: a few ids)
    req = torch.from_numpy(np.concatenate([np.asarray(x, np.int64) for x in need]) if sum(len(x) for x in need) else np.zeros(0, np.int64)).to(device)
    got, got_counts = _all_to_all_tensor(dist, torch, device, req, [len(x) for x in need])
    got = got.cpu().numpy(); o = np.concatenate([[0], np.cumsum(got_counts)]).astype(np.int64)
    wanted = [got[o[r]:o[r + 1]] - base for r in range(world)]                               # local indices, ascending
    # my arrays, exported once into tensors
    meta = ss_local.export_meta(); P, M, _ = ss_local.totals()
    a32 = torch.zeros(max(3 * P, 1), dtype=torch.int32, device=device); m64 = torch.zeros(max(M, 1), dtype=torch.int64, device=device)
    if on_dev:
        torch.cuda.synchronize(device)
        ss_local.export_arrays(seed=a32.data_ptr(), pos=a32.data_ptr() + 4 * P, ctgcanon=a32.data_ptr() + 8 * P, markers=m64.data_ptr(), device=True)
    else:
        v = a32.numpy().view(np.uint32)
        ss_local.export_arrays(seed=v[0:P], pos=v[P:2 * P], ctgcanon=v[2 * P:3 * P], markers=m64.numpy().view(np.uint64))
    po, mo, co = (meta[k].astype(np.int64) for k in ("pos_off", "marker_off", "contig_off"))
    s32, s64, hdr, c32, c64, ch = [], [], [], [], [], []
    for r in range(world):
        loc = wanted[r]
        h = [np.array([len(loc)], np.int64)]
        for g in loc:
            h.append(np.array([po[g + 1] - po[g], mo[g + 1] - mo[g], co[g + 1] - co[g], meta["total_len"][g]], np.int64))
            h.append(meta["contig_lengths"][co[g]:co[g + 1]].astype(np.int64))
        h = np.concatenate(h); hdr.append(h); ch.append(len(h))
        parts = [a32[k * P + po[g]:k * P + po[g + 1]] for k in range(3) for g in loc]       # all seeds, then all positions, then all contig/strand words
        s32.append(torch.cat(parts) if parts else a32[:0]); c32.append(int(s32[-1].numel()))
        parts = [m64[mo[g]:mo[g + 1]] for g in loc]
        s64.append(torch.cat(parts) if parts else m64[:0]); c64.append(int(s64[-1].numel()))
    r32, n32 = _all_to_all_tensor(dist, torch, device, torch.cat(s32), c32)
    r64, n64 = _all_to_all_tensor(dist, torch, device, torch.cat(s64), c64)
    rh, nh = _all_to_all_tensor(dist, torch, device, torch.from_numpy(np.concatenate(hdr)).to(device), ch)
    rh = rh.cpu().numpy()
    # assemble one flat set of all received genomes: sources in rank order = increasing global id
    ids, npos, nmk, nct, tl, cl = [], [], [], [], [], []
    oh = 0
    for r in range(world):
        h = rh[oh:oh + nh[r]]; oh += nh[r]
        if not len(h):
            continue
        k = int(h[0]); x = 1
        assert k == len(need[r])
        for g in range(k):
            npos.append(int(h[x])); nmk.append(int(h[x + 1])); nct.append(int(h[x + 2])); tl.append(int(h[x + 3])); x += 4
            cl.append(h[x:x + nct[-1]]); x += nct[-1]
        ids.extend(int(v) for v in need[r])
    o32 = np.concatenate([[0], np.cumsum(n32)]).astype(np.int64)
    seeds, poss, ccs = [], [], []
    for r in range(world):
        blk = r32[o32[r]:o32[r + 1]]; t = blk.numel() // 3
        seeds.append(blk[0:t]); poss.append(blk[t:2 * t]); ccs.append(blk[2 * t:3 * t])
    seed_t, pos_t, cc_t = (torch.cat(x).contiguous() if x else a32[:0] for x in (seeds, poss, ccs))
    cum = lambda v: np.concatenate([[0], np.cumsum(v)]).astype(np.uint64)
    gmeta = dict(pos_off=cum(npos), marker_off=cum(nmk), contig_off=cum(nct), contig_lengths=(np.concatenate(cl) if cl else np.zeros(0, np.int64)).astype(np.uint32),
                 total_len=np.array(tl, np.uint64), genome_rank=np.array(ids, np.uint32))
    if on_dev:
        torch.cuda.synchronize(device)
        keep = [t_ if t_.numel() else torch.zeros(1, dtype=t_.dtype, device=device) for t_ in (seed_t, pos_t, cc_t, r64)]
        ss = ctx.import_flat(params, gmeta, seed=keep[0].data_ptr(), pos=keep[1].data_ptr(), ctgcanon=keep[2].data_ptr(), markers=keep[3].data_ptr(), device=True)
    else:
        ss = ctx.import_flat(params, gmeta, seed=seed_t.numpy().view(np.uint32), pos=pos_t.numpy().view(np.uint32), ctgcanon=cc_t.numpy().view(np.uint32),
                             markers=r64.numpy().view(np.uint64))
    return ss, np.array(ids, np.uint32)


def _gather_markers(ctx, ss_local, params, dist, rank, world, torch, device):
    """All-gather of the marker sets: returns a markers-only SketchSet of ALL genomes (genome order = rank order) for the
    screen.  On GPUs the markers stay in device memory end to end (export -> RCCL all_gather -> import).  Contig tables are
    not needed for screening: every genome is entered as one contig of its total length."""
    n_local = len(ss_local)
    meta = ss_local.export_meta()
    _, M, _ = ss_local.totals()
    on_dev = device.type == "cuda"
    mine = torch.from_numpy(np.concatenate([np.diff(meta["marker_off"]).astype(np.int64), meta["total_len"].astype(np.int64)])).to(device)
    allmeta = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allmeta, mine)                                                   # collective 1: marker counts + total lengths
    allmeta = [m.cpu().numpy() for m in allmeta]
    counts = np.concatenate([m[:n_local] for m in allmeta]).astype(np.uint64); total_len = np.concatenate([m[n_local:] for m in allmeta]).astype(np.uint64)
    Ms = [int(m[:n_local].sum()) for m in allmeta]
    mk = torch.zeros(max(M, 1), dtype=torch.int64, device=device)                   # u64 bit patterns
    if M:
        if on_dev:
            torch.cuda.synchronize(device)                                          # the library copies on its own stream: torch's fill must have landed
            ss_local.export_arrays(markers=mk.data_ptr(), device=True)
        else:
            ss_local.export_arrays(markers=mk.numpy().view(np.uint64))
    parts = _all_gather_padded(dist, torch, device, mk[:M], Ms)                      # collective 2: the markers
    allmk = torch.cat(parts) if sum(Ms) else torch.zeros(1, dtype=torch.int64, device=device)
    n_total = n_local * world
    mo = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    gmeta = dict(pos_off=np.zeros(n_total + 1, np.uint64), marker_off=mo, contig_off=np.arange(n_total + 1, dtype=np.uint64),
                 contig_lengths=np.minimum(total_len, np.uint64(0x7FFF0000)).astype(np.uint32), total_len=total_len, genome_rank=np.arange(n_total, dtype=np.uint32))
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
    # 1. markers of every genome, everywhere; this rank screens its rows (local genomes i) against all later genomes j > i:
    #    an n_local x N block of the triangle's screen (triangle.rs:55-90)
    markers_only, _keep = _gather_markers(ctx, ss_local, params, dist, rank, world, torch, device)
    gi, gj = ctx.screen_rows(markers_only, base, n_local, identity, rescue_small)
    markers_only.close()
    owner_j = gj // n_local
    # 2. sketches of remote partners j of my rows i -- only if some pair, anywhere, crosses blocks
    need = [np.unique(gj[owner_j == r]) if r != rank else np.zeros(0, np.uint32) for r in range(world)]
    crossing = torch.tensor([sum(len(x) for x in need)], dtype=torch.int64, device=device)
    dist.all_reduce(crossing)                                                        # collective 3
    ss_rem, rem_ids = None, np.zeros(0, np.uint32)
    if int(crossing.item()) > 0:
        ss_rem, rem_ids = _exchange_sketches(ctx, ss_local, params, need, base, dist, rank, world, torch, device)
    rem_index = {int(g): k for k, g in enumerate(rem_ids)}
    local_pair = owner_j == rank
    res_parts = []
    n_chained = int(len(gi))
    if local_pair.any():
        r = ctx.chain_pairs(ss_local, None, gi[local_pair] - base, gj[local_pair] - base, map_params)
        res_parts.append((gi[local_pair], gj[local_pair], r))
    if (~local_pair).any():
        qi = np.array([rem_index[int(g)] for g in gj[~local_pair]], dtype=np.uint32)
        r = ctx.chain_pairs(ss_local, ss_rem, gi[~local_pair] - base, qi, map_params)      # ref = genome i (local), query = genome j (remote)
        res_parts.append((gi[~local_pair], gj[~local_pair], r))
    if ss_rem is not None:
        ss_rem.close()
    if res_parts:
        ai = np.concatenate([p[0] for p in res_parts]); aj = np.concatenate([p[1] for p in res_parts]); ar = np.concatenate([p[2] for p in res_parts])
        keep = ar["ani"] > 0.1                                                       # triangle.rs:99
        ai, aj, ar = ai[keep], aj[keep], ar[keep]
    else:
        ai = np.zeros(0, np.uint32); aj = np.zeros(0, np.uint32); ar = np.zeros(0, B.RESULT_DTYPE)
    # 3. results: (i, j, record) rows as bytes, all-gathered
    rec = np.zeros(len(ai), np.dtype([("i", np.uint32), ("j", np.uint32), ("r", B.RESULT_DTYPE)]))
    rec["i"] = ai; rec["j"] = aj; rec["r"] = ar
    counts = torch.tensor([rec.nbytes, n_chained], dtype=torch.int64, device=device)
    allc = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts)                                                    # collective 4
    allc = [c.cpu().numpy() for c in allc]
    payload = torch.from_numpy(np.frombuffer(rec.tobytes(), np.uint8).copy()).to(device) if rec.nbytes else torch.zeros(0, dtype=torch.uint8, device=device)
    parts = _all_gather_padded(dist, torch, device, payload, [int(c[0]) for c in allc])    # collective 5
    n_total_chained = int(sum(int(c[1]) for c in allc))
    if rank != 0:
        return None, None, None, n_total_chained
    allrec = np.concatenate([np.frombuffer(p.cpu().numpy().tobytes(), rec.dtype) for p in parts])
    order = np.lexsort((allrec["j"], allrec["i"]))
    allrec = allrec[order]
    return allrec["i"].astype(np.uint32), allrec["j"].astype(np.uint32), allrec["r"].copy(), n_total_chained
