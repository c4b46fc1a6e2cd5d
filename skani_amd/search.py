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
           learned_ani=None, compute_ci=True, ani_min=0.5):
    """Returns (query_idx, ref_idx, results) sorted by query then ANI descending, at most n_max rows per query."""
    c = db.shards[0].params.c if db.shards else 125
    if learned_ani is None:
        learned_ani = use_learned_ani(c, False, False, median)                   # search.rs:53
    if use_index is None:
        use_index = (n_query_files if n_query_files is not None else len(queries)) > FULL_INDEX_THRESH    # parse.rs:960
    mp = MapParams(min_af=min_af, robust=robust, median=median, learned_ani=learned_ani, compute_ci=compute_ci)
    qs, rs, res = [], [], []
    if not db.shards:
        return np.zeros(0, np.uint32), np.zeros(0, np.int64), np.zeros(0, B.RESULT_DTYPE)
    # one screen of all queries against the whole database's markers, then chaining shard by shard
    q_all, r_all = ctx.screen(db.marker_index(ctx), queries, screen_val, SCREEN_REFS_INDICES if use_index else SCREEN_QUICK, False)
    if len(q_all) == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.int64), np.zeros(0, B.RESULT_DTYPE)
    shard_of = (np.searchsorted(db.offsets, r_all, side="right") - 1).astype(np.uint32)
    local = (r_all.astype(np.int64) - db.offsets[shard_of]).astype(np.uint32)
    out = ctx.chain_pairs_multi(db.shards, queries, shard_of, local, q_all, mp)    # chain_seeds(ref_sketch, query_sketch): search.rs:175
    keep = out["ani"] > ani_min                                                    # search.rs:176
    qs.append(q_all[keep]); rs.append(r_all[keep].astype(np.int64)); res.append(out[keep])
    if not qs:
        return np.zeros(0, np.uint32), np.zeros(0, np.int64), np.zeros(0, B.RESULT_DTYPE)
    q = np.concatenate(qs); r = np.concatenate(rs); o = np.concatenate(res)
    order = np.lexsort((r, -o["ani"].astype(np.float64), q))
    q, r, o = q[order], r[order], o[order]
    if n_max is not None:
        rank = np.arange(len(q)) - np.searchsorted(q, q, side="left")
        sel = rank < n_max
        q, r, o = q[sel], r[sel], o[sel]
    return q, r, o
