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

    def close(self):
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
        shards.append(ctx.sketch_records(genomes[a:a + shard_genomes], params, nm))
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
    for k, shard in enumerate(db.shards):
        q, r = ctx.screen(shard, queries, screen_val, SCREEN_REFS_INDICES if use_index else SCREEN_QUICK, False)
        if len(q) == 0:
            continue
        out = ctx.chain_pairs(shard, queries, r, q, mp)                            # chain_seeds(ref_sketch, query_sketch): search.rs:175
        keep = out["ani"] > ani_min                                                # search.rs:176
        qs.append(q[keep]); rs.append(r[keep].astype(np.int64) + db.offsets[k]); res.append(out[keep])
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
