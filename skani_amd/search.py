"""`skani search` body (search.rs:97-282) with the database RESIDENT in HBM instead of lazily deserialised per hit.

A database is one or more SketchSets ("shards"; one sketch-table build handles < 2^32 seed positions, e.g. ~60k genomes of
5 Mbp at c=70).  Per shard: marker screen of all (query, ref) pairs -- check_markers_quickly with rescue_small=false
(search.rs:127) for <= 50 query files, screen_refs_indices otherwise (search.rs:134, parse.rs:960) -- then
chain_seeds(ref, query) for every passing pair, keep ani > 0.5 (search.rs:176), best `n` hits per query
(write_query_ref_list, file_io.rs:640-655)."""
import numpy as np

from . import _binding as B
from .api import use_learned_ani, MapParams

FULL_INDEX_THRESH = 50     # params.rs:50
SCREEN_QUICK, SCREEN_REFS_INDICES = 1, 2


class SketchDB:
    def __init__(self, shards, names=None):
        self.shards = list(shards)
        self.offsets = np.concatenate([[0], np.cumsum([len(s) for s in self.shards])]).astype(np.int64)
        self.names = names

    def __len__(self):
        return int(self.offsets[-1])

    def marker_index(self, ctx):
        """Markers-only sketch set over ALL genomes of the database (global genome order), screened with one call; the library
        caches its sorted (marker, genome) incidence list inside the set, so the index is built once per database
        (markers.bin's role in search.rs:37-60)."""
        if getattr(self, "_marker_index", None) is None:
            if len(self.shards) == 1:
                self._marker_index = self.shards[0]
            else:
                metas = [s.export_meta() for s in self.shards]
                mks = []
                for s in self.shards:
                    _, M, _ = s.totals(); a = np.zeros(M, np.uint64); s.export_arrays(markers=a); mks.append(a)
                n = len(self)
                cat = lambda key: np.concatenate([[0]] + [m[key][1:] + off for m, off in zip(metas, np.cumsum([0] + [int(m[key][-1]) for m in metas[:-1]]))]).astype(np.uint64)
                meta = dict(pos_off=np.zeros(n + 1, np.uint64), marker_off=cat("marker_off"), contig_off=cat("contig_off"),
                            contig_lengths=np.concatenate([m["contig_lengths"] for m in metas]).astype(np.uint32),
                            total_len=np.concatenate([m["total_len"] for m in metas]).astype(np.uint64), genome_rank=np.arange(n, dtype=np.uint32))
                self._marker_index = ctx.import_flat(self.shards[0].params, meta, markers=np.concatenate(mks))
        return self._marker_index

    def close(self):
        mi = getattr(self, "_marker_index", None)
        if mi is not None and all(mi is not s for s in self.shards):
            mi.close()
        for s in self.shards:
            s.close()


def open_db(ctx, folder, seeding_mode=1):
    """A folder written by `skani sketch` / `skani-hip sketch` / save_database, made resident as one shard (search.rs:31-100)."""
    from .formats import load_database
    ss, infos = load_database(ctx, folder, seeding_mode)
    db = SketchDB([ss], [i["file_name"] for i in infos]); db.infos = infos
    return db


def build_db(ctx, genomes, params, names=None, shard_genomes=None):
    """genomes: list of lists of (name, seq) records.  shard_genomes bounds the genomes per shard (None = by size)."""
    if shard_genomes is None:
        per_genome = max(1, sum(len(s) for _, s in genomes[0]) // params.c) if genomes else 1
        shard_genomes = max(1, (3 << 30) // per_genome)
    shards = []
    for a in range(0, len(genomes), shard_genomes):
        nm = names[a:a + shard_genomes] if names is not None else None
        shards.append(ctx.sketch_records(genomes[a:a + shard_genomes], params, nm, compact=True))   # resident: SKH_SKETCH_COMPACT
    return SketchDB(shards, names)


def search(ctx, db, queries, screen_val=0.0, n_query_files=None, use_index=None, n_max=10000000, min_af=-1.0, robust=False, median=False,
           learned_ani=None, compute_ci=True, ani_min=0.5, host_times=None):
    """Returns (query_idx, ref_idx, results) sorted by query then ANI descending, at most n_max rows per query.
    host_times: a dict that collects the wall time of the call's stages as the host sees them (seconds, added up over calls)."""
    import time
    t_last = [time.perf_counter()]
    def lap(name):
        if host_times is not None:
            now = time.perf_counter(); host_times[name] = host_times.get(name, 0.0) + now - t_last[0]; t_last[0] = now
    c = db.shards[0].params.c if db.shards else 125
    if learned_ani is None:
        learned_ani = use_learned_ani(c, False, False, median)                   # search.rs:53
    if use_index is None:
        use_index = (n_query_files if n_query_files is not None else len(queries)) > FULL_INDEX_THRESH    # parse.rs:960
    mp = MapParams(min_af=min_af, robust=robust, median=median, learned_ani=learned_ani, compute_ci=compute_ci)
    if not db.shards:
        return np.zeros(0, np.uint32), np.zeros(0, np.int64), np.zeros(0, B.RESULT_DTYPE)
    # one screen of all queries against the whole database's markers, then chaining shard by shard
    lap("before the screen")
    q_all, r_all = ctx.screen(db.marker_index(ctx), queries, screen_val, SCREEN_REFS_INDICES if use_index else SCREEN_QUICK, False)
    lap("screen call")
    if len(q_all) == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.int64), np.zeros(0, B.RESULT_DTYPE)
    shard_of = (np.searchsorted(db.offsets, r_all, side="right") - 1).astype(np.uint32)
    local = (r_all.astype(np.int64) - db.offsets[shard_of]).astype(np.uint32)
    lap("shard of every hit")
    out = ctx.chain_pairs_multi(db.shards, queries, shard_of, local, q_all, mp)    # chain_seeds(ref_sketch, query_sketch): search.rs:175
    lap("chain call")
    # keep ani > 0.5 (search.rs:176), rows by query, then ANI descending, then reference (file_io.rs:640-655), at most n_max per query.  One stable sort on
    # (query, ~ANI bits): a positive float32 orders like its bit pattern, and ties keep the reference-ascending order the screen returned (re-made if it is not there).
    # np.take, not out[idx]: fancy indexing of a structured array goes field by field (2.4 ms for 20,000 rows of 64 B where take needs 0.15).
    ani = np.ascontiguousarray(out["ani"])
    keep = np.flatnonzero(ani > ani_min)
    q = q_all[keep]; r = r_all[keep].astype(np.int64)
    key = (q.astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - ani[keep].view(np.uint32).astype(np.uint64))
    if len(q) > 1 and not ((q[1:] > q[:-1]) | ((q[1:] == q[:-1]) & (r[1:] >= r[:-1]))).all():
        pre = np.lexsort((r, q)); keep = keep[pre]; q = q[pre]; r = r[pre]; key = key[pre]
    order = np.argsort(key, kind="stable")
    q = q[order]; r = r[order]; idx = keep[order]
    if n_max is not None and n_max < len(q):
        rank = np.arange(len(q)) - np.searchsorted(q, q, side="left")
        sel = rank < n_max
        q, r, idx = q[sel], r[sel], idx[sel]
    o = np.take(out, idx)
    lap("rows kept, sorted, cut")
    return q, r, o
