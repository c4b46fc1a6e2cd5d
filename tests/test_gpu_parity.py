"""GPU parity tests proper: the HIP path (libskani_hip.so through the C ABI) against the CPU oracle and the
reference's golden vectors, on an MI355X.  `python -m pytest tests -m gpu`."""
import os

import numpy as np
import pytest

import skani_amd as sk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests import parity_cases as pc
from tests.helpers import mutate, ora, random_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = sk.Context(0)          # raises loudly when libskani_hip.so or the GPU is missing
    yield c
    c.close()


def test_library_is_the_hip_build(ctx):
    import os
    assert os.path.exists(sk.library_path())
    maps = open("/proc/self/maps").read()
    assert "libskani_hip.so" in maps and "libskani_emu" not in maps


def test_seeding_golden(ctx): pc.case_seeding_golden_plasmid(ctx)
def test_seeding_fixtures(ctx): pc.case_seeding_fixtures(ctx)
def test_seeding_ecoli_w(ctx): pc.case_seeding_ecoli_w(ctx)
def test_pack_every_byte(ctx): pc.case_pack_every_byte(ctx)
def test_pack_in_batches(ctx): pc.case_pack_in_batches(ctx)
def test_seeding_low_complexity(ctx): pc.case_seeding_low_complexity(ctx)
def test_pinned_triples(ctx): pc.case_pinned_triples(ctx)
def test_w_vs_w(ctx): pc.case_w_vs_w(ctx)
def test_viruses(ctx): pc.case_viruses_individual(ctx)
def test_triangle_synthetic(ctx): pc.case_triangle_synthetic(ctx)
def test_triangle_dense_sketches(ctx): pc.case_triangle_synthetic(ctx, params=((1, 8), (0, 3)), length=60000)   # c < 10: chain band beyond 256 anchors
def test_screen_rules(ctx): pc.case_screen_rules(ctx)
def test_screen_marker_prefix_groups(ctx): pc.case_screen_marker_prefix_groups(ctx)
def test_marker_set_sizes(ctx): pc.case_marker_set_sizes(ctx)
def test_every_genome_resalted(monkeypatch):
    """SKH_TUNE_BUILD_RESALT_ALL=1 (tests/test_emu_pipeline.py has the why): the second salt through the sketch call, whose per-genome tables are made ahead of the build's end."""
    monkeypatch.setenv("SKH_TUNE_BUILD_RESALT_ALL", "1")
    c = sk.Context(0)
    try:
        pc.case_triangle_synthetic(c)
    finally:
        c.close()


@pytest.mark.parametrize("env", [{"SKH_TUNE_SKEYS_AVG": "16"}, {"SKH_TUNE_SKEYS_AVG": "16", "SKH_TUNE_SKEYS_CAP": "16"}, {"SKH_TUNE_SCREEN_SORT_RADIX": "1"},
                                 {"SKH_TUNE_SCREEN_COL_ORDER": "0"}, {"SKH_TUNE_SCREEN_COL_ORDER": "2"}, {"SKH_TUNE_SCREEN_COUNT_ROWS": "0"}, {"SKH_TUNE_MARKER_GATE": "0"}])   # (the last three: the count kernel without its column order / with it also in small key-range parts / the count kernel of rounds 1-5; then the sketch call without the marker sets' head start)
def test_incidence_sort_through_the_sketch_call(monkeypatch, env):
    """tests/test_emu_pipeline.py has the why: the sketch call's two-halves way through the incidence sort with tiny buckets, with the radix-sort way out, with the radix sort alone."""
    for k, v in env.items(): monkeypatch.setenv(k, v)
    c = sk.Context(0)
    try:
        pc.case_triangle_synthetic(c)
        pc.case_screen_rules(c)
    finally:
        c.close()


def test_screen_incidence_sort():
    def make_ctx(env):
        for k, v in env.items(): os.environ[k] = v
        try: return sk.Context(0)
        finally:
            for k in env: os.environ.pop(k, None)
    pc.case_screen_incidence_sort(make_ctx)
def test_screen_count_walks():
    def make_ctx(env):
        for k, v in env.items(): os.environ[k] = v
        try: return sk.Context(0)
        finally:
            for k in env: os.environ.pop(k, None)
    pc.case_screen_count_walks(make_ctx)
def test_screen_from_cells_large_rows(ctx):
    pc.case_screen_from_cells_large_rows(ctx)                # beyond 16,384 genomes: an LDS row of more than 64 KB; columns in collection order
    pc.case_screen_from_cells_large_rows(ctx, N=6000)        # 4,097 .. 16,384 genomes: the column order's labels through the radix sort
def test_degenerate(ctx): pc.case_degenerate_pairs(ctx)
def test_fragmented_genomes(ctx): pc.case_fragmented_genomes(ctx)
def test_database_formats(ctx, tmp_path): pc.case_database_formats(ctx, str(tmp_path))


def test_pack_device_buffer_at_odd_addresses(ctx):
    """skh_genomes_pack with bases already in HBM (what bench.py does) at byte addresses 1, 2 and 3 past a word boundary and with the buffer ending
    right behind the last base: the kernel's aligned four-word loads must neither shift the bases nor read past the buffer's last word."""
    import torch
    recs = [("a", random_genome(70001, 41, 0.0005)), ("b", random_genome(1203, 42)), ("c", random_genome(33333, 43).lower())]
    flat = np.frombuffer(b"".join(s for _, s in recs), np.uint8)
    off = np.zeros(len(recs) + 1, np.uint64); off[1:] = np.cumsum([len(s) for _, s in recs])
    want = ora.sketch_records(recs, 30, 15, 200, "x", 1)
    for shift in (0, 1, 2, 3):
        t = torch.zeros(len(flat) + shift, dtype=torch.uint8, device="cuda:0")
        t[shift:] = torch.from_numpy(flat.copy()).to("cuda:0")
        torch.cuda.synchronize()
        gs = ctx.pack_buffer(None, off, np.zeros(len(recs), np.uint32), 1, sk.SEED_AVX2, device_ptr=t.data_ptr() + shift)
        ss = ctx.sketch_genomes(gs, sk.SketchParams(30, 15, 200, sk.SEED_AVX2))
        pc.assert_sketch_equal(ss, 0, want)
        ss.close(); gs.close()


def test_w_derivatives_triangle(ctx):
    """SURVEY 8d config 2 substitute: E. coli W + 5 substitution derivatives (0.5/1/2/4/8 %), 15 pairs, default
    -c 125 -k 15 -m 1000, learned ANI on; GPU vs oracle on every field."""
    from tests.helpers import MODEL_C125, golden_records
    W = golden_records("e.coli-W.fasta.gz")
    genomes = [W] + [[(W[0][0], mutate(W[0][1], r, 0x5EED0001 + i))] for i, r in enumerate((0.005, 0.01, 0.02, 0.04, 0.08))]
    names = ["w%d.fa" % i for i in range(len(genomes))]
    ss = ctx.sketch_records(genomes, sk.SketchParams(), names)
    osk = [ora.sketch_records(g, file_name=names[i]) for i, g in enumerate(genomes)]
    for g in range(len(genomes)):
        pc.assert_sketch_equal(ss, g, osk[g])
    i, j, res, nch = ctx.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
    oi, oj, ores, onch, _ = ora.triangle(osk, model=ora.Model(MODEL_C125))
    assert nch == onch == 15 and np.array_equal(i, oi) and np.array_equal(j, oj)
    for x in range(len(res)):
        pc.assert_result_close(res[x], ores[x], (int(i[x]), int(j[x])))
    # size-independent properties: symmetry of roles (chain(a,b) vs chain(b,a) swap the aligned fractions), self = 1
    fwd = ctx.chain_pairs(ss, None, [0, 1], [3, 4], sk.MapParams())
    rev = ctx.chain_pairs(ss, None, [3, 4], [0, 1], sk.MapParams())
    assert np.allclose(fwd["ani"], rev["ani"], atol=2e-3)
    selfr = ctx.chain_pairs(ss, None, list(range(6)), list(range(6)), sk.MapParams())
    assert (selfr["ani"] >= 0.9999).all() and (selfr["af_ref"] >= 0.99).all()


def test_many_genomes_batching(ctx):
    """Several hundred small genomes: exercises multi-tile launches, the screen matrix, pair batching and ordering."""
    rng = np.random.default_rng(3)
    genomes, names = [], []
    for cl in range(12):
        root = random_genome(int(rng.integers(30000, 60000)), 1000 + cl)
        for m in range(6):
            genomes.append([("c", mutate(root, float(rng.uniform(0.003, 0.06)), cl * 10 + m))]); names.append("b%03d.fa" % len(names))
    ss = ctx.sketch_records(genomes, sk.SketchParams(c=30, marker_c=200), names)
    osk = [ora.sketch_records(g, 30, 15, 200, names[i]) for i, g in enumerate(genomes)]
    i, j, res, nch = ctx.triangle(ss, sk.MapParams(compute_ci=True))
    oi, oj, ores, onch, _ = ora.triangle(osk)
    assert nch == onch and np.array_equal(i, oi) and np.array_equal(j, oj) and len(i) >= 12 * 15
    for x in range(len(res)):
        pc.assert_result_close(res[x], ores[x], (int(i[x]), int(j[x])))
    # sharded triangle (the multi-GPU partition): union of parts == whole
    parts = [ctx.triangle(ss, sk.MapParams(compute_ci=True), part=r, n_parts=3) for r in range(3)]
    allp = sorted((int(a), int(b)) for pi, pj, _, _ in parts for a, b in zip(pi, pj))
    assert allp == sorted(zip(i.tolist(), j.tolist()))


def test_full_size_genomes_properties_and_oracle_sample(ctx):
    """BASELINE-sized genomes (~5 Mbp, generated on the GPU like bench.py): size-independent properties on 100 genomes
    (every within-clade pair chained and kept, nothing across clades, ANI decreasing with divergence, self-consistency
    of the sharded run) plus field-by-field comparison with the oracle on one clade (190 pairs)."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    bases, coff, cgen, ng, host = bench.make_genomes(torch, dev, np.arange(100), keep_host=True)
    host = host[:20]
    torch.cuda.synchronize()
    gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
    assert gs.total_bases == int(coff[-1])
    ss = ctx.sketch_genomes(gs, sk.SketchParams(), genome_rank=np.arange(ng, dtype=np.uint32))
    del bases
    mp = sk.MapParams(learned_ani=True, compute_ci=True)
    i, j, res, nch = ctx.triangle(ss, mp)
    assert nch == 5 * 190 and len(i) == 5 * 190
    assert ((i // 20) == (j // 20)).all()                       # only within-clade pairs survive the screen
    assert (res["ani"] > 0.80).all() and (res["ani"] <= 1.0).all() and (res["af_ref"] > 0.25).all()
    # sharded == whole
    parts = [ctx.triangle(ss, mp, part=r, n_parts=4) for r in range(4)]
    got = sorted((int(a), int(b), float(r["ani"])) for pi, pj, pr, _ in parts for a, b, r in zip(pi, pj, pr))
    assert got == sorted((int(a), int(b), float(r["ani"])) for a, b, r in zip(i, j, res))
    # oracle on clade 0 (host copies of the same bytes)
    from tests.helpers import MODEL_C125
    names = ["s%04d.fa" % k for k in range(20)]
    osk = [ora.sketch_records(g, file_name=names[k]) for k, g in enumerate(host)]
    for k in (0, 7, 19):
        pc.assert_sketch_equal(ss, k, osk[k])
    oi, oj, ores, onch, _ = ora.triangle(osk, model=ora.Model(MODEL_C125))
    sel = (i < 20) & (j < 20)
    assert np.array_equal(i[sel], oi) and np.array_equal(j[sel], oj)
    for x, y in zip(res[sel], ores):
        pc.assert_result_close(x, y)


def test_config3_full_size(ctx):
    """BASELINE config 3 at its full size: the triangle over 1,000 synthetic ~5 Mbp genomes (what bench.py times).  Size-independent
    properties on the whole result -- 9,500 pairs pass the screen, all inside clades, all kept -- plus field-by-field comparison with the
    oracle on one clade from the middle of the collection (190 pairs, ~2 s of CPU)."""
    import torch
    import bench
    from tests.helpers import MODEL_C125
    dev = torch.device("cuda", 0)
    bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, np.arange(1000))
    torch.cuda.synchronize()
    gs = ctx.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
    del bases
    ss = ctx.sketch_genomes(gs, sk.SketchParams(), genome_rank=np.arange(ng, dtype=np.uint32))
    gs.close()
    i, j, res, nch = ctx.triangle(ss, sk.MapParams(learned_ani=True, compute_ci=True))
    assert nch == 9500 and len(i) == 9500 and ((i // 20) == (j // 20)).all() and (i < j).all()
    assert (res["ani"] > 0.80).all() and (res["ani"] <= 1.0).all() and (res["ci_lower"] <= res["ci_upper"]).all()
    clade = 23
    _, _, _, _, host = bench.make_genomes(torch, dev, np.arange(clade * 20, clade * 20 + 20), keep_host=True)
    osk = [ora.sketch_records(g, file_name="s%05d.fa" % (clade * 20 + k)) for k, g in enumerate(host)]
    for k in (0, 11):
        pc.assert_sketch_equal(ss, clade * 20 + k, osk[k])
    oi, oj, ores, onch, _ = ora.triangle(osk, model=ora.Model(MODEL_C125))
    sel = (i // 20) == clade
    assert np.array_equal(i[sel] - clade * 20, oi) and np.array_equal(j[sel] - clade * 20, oj)
    for x, y in zip(res[sel], ores):
        pc.assert_result_close(x, y)
    ss.close()
    torch.cuda.empty_cache()


def test_config5_full_size():
    """BASELINE config 5 at its full size: 1,000 queries against a 65,000-genome database (c = 70, --medium) resident in HBM (~165 GB;
    building it takes ~20 s).  Properties: every query finds exactly the 20 members of its own clade and nothing else; the results of 20 sampled
    (query, hit) pairs are compared field by field with the oracle, which sketches the same bytes on the host."""
    import argparse
    import torch
    import bench
    from tests.helpers import MODEL_C125
    dev = torch.device("cuda", 0)
    c = sk.Context(0)
    try:
        bench.C, bench.CLADE = 70, 20
        args = argparse.Namespace(c=70, db_genomes=65000, queries=1000, mean_len=5_000_000, steps=1, warmup=0)
        line, (q, r, res, qclades) = bench.run_search(args, torch, sk, c, dev)
        assert line["config"]["hits"] == 20000 and line["config"]["hits_in_own_clade"] == 20000 and len(q) == 20000
        assert np.array_equal(np.bincount(q, minlength=1000), np.full(1000, 20)) and ((r // 20) == qclades[q]).all()
        assert (res["ani"] > 0.8).all()
        rng = np.random.default_rng(5)
        model = ora.Model(MODEL_C125)                                             # c = 70: |70 - 125| < |70 - 200| (regression.rs:15-22)
        for x in rng.choice(len(q), 20, replace=False):
            _, _, _, _, hr = bench.make_genomes(torch, dev, [int(r[x])], keep_host=True)
            qb, qoff, _, _ = bench.make_queries(torch, dev, [int(qclades[q[x]])], first=int(q[x]))
            qh = qb.cpu().numpy()
            oref = ora.sketch_records(hr[0], 70, 15, 1000, "s%07d.fa" % int(r[x]))
            oq = ora.sketch_records([("q", qh)], 70, 15, 1000, "t%07d.fa" % int(q[x]))           # query names sort after every database name
            want = ora.chain_seeds(oref, oq, model=model)
            pc.assert_result_close(res[x], want, (int(q[x]), int(r[x])))
    finally:
        bench.C, bench.CLADE = 125, 20
        c.close()
        torch.cuda.empty_cache()


def test_search_resident_db(ctx): pc.case_search_resident_db(ctx)
def test_large_pair(ctx): pc.case_large_pair(ctx)
def test_edge_cases(ctx): pc.case_edge_cases_and_errors(ctx)


def test_flat_export_import_device_roundtrip(ctx):
    """The multi-GPU exchange format: whole-set export into torch device tensors and import from device pointers gives back
    the same sketches (and a markers-only set screens identically)."""
    import torch
    genomes = pc.synthetic_clades(n_clades=2, members=3, length=80000, seed=77)
    names = ["f%02d.fa" % i for i in range(len(genomes))]
    ss = ctx.sketch_records(genomes, sk.SketchParams(), names)
    P, M, NC = ss.totals(); meta = ss.export_meta()
    dev = torch.device("cuda", 0)
    t = {k: torch.zeros(max(P, 1), dtype=torch.int32, device=dev) for k in ("seed", "pos", "cc")}
    mk = torch.zeros(max(M, 1), dtype=torch.int64, device=dev)
    ss.export_arrays(seed=t["seed"].data_ptr(), pos=t["pos"].data_ptr(), ctgcanon=t["cc"].data_ptr(), markers=mk.data_ptr(), device=True)
    torch.cuda.synchronize()
    ss2 = ctx.import_flat(sk.SketchParams(), meta, seed=t["seed"].data_ptr(), pos=t["pos"].data_ptr(), ctgcanon=t["cc"].data_ptr(), markers=mk.data_ptr(),
                          device=True, names=names)
    for g in range(len(genomes)):
        a, b = ss.export(g), ss2.export(g)
        for k in ("seed", "pos", "ctgcanon", "markers", "contig_lengths"):
            assert np.array_equal(a[k], b[k]), (g, k)
    mp = sk.MapParams(learned_ani=True, compute_ci=True)
    r1 = ctx.triangle(ss, mp); r2 = ctx.triangle(ss2, mp)
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1]) and r1[2].tobytes() == r2[2].tobytes()
    meta0 = dict(meta); meta0["pos_off"] = np.zeros_like(meta["pos_off"])
    mo = ctx.import_flat(sk.SketchParams(), meta0, markers=mk.data_ptr(), device=True)
    a1 = ctx.screen(ss, None); a2 = ctx.screen(mo, None)
    assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1])


def test_small_budgets_force_multi_batch_paths(monkeypatch):
    monkeypatch.setenv("SKH_TUNE_SEED_SCRATCH_BYTES", "6000")
    monkeypatch.setenv("SKH_TUNE_SCREEN_CELLS", "20")
    monkeypatch.setenv("SKH_TUNE_CHAIN_ANCHORS", "3000")
    monkeypatch.setenv("SKH_TUNE_CHAIN_SUPER_TILES", "2")
    monkeypatch.setenv("SKH_TUNE_CHAIN_DP_LDS_SLOTS", "1")
    monkeypatch.setenv("SKH_TUNE_JOIN_BITMAP_WORDS", "8")                # genomes with more than 256 buckets probe without the staged bitmap
    monkeypatch.setenv("SKH_TUNE_BUILD_SLICE_MAX", "1")                 # genomes with more than one table slice: no slice lists, the slices re-scan the genome
    monkeypatch.setenv("SKH_TUNE_MARKER_LDS_MAX", "40")                 # genomes with more than 40 raw markers: marker sets by the device-wide passes
    monkeypatch.setenv("SKH_TUNE_BUILD_MATCH_CAP", "64")                # table slices with more than 64 positions re-scan the genome instead of listing them in LDS
    monkeypatch.setenv("SKH_TUNE_GREEDY_LEN_LIMIT", "3000")             # pairs with a chain interval of 3 kb or more are handed from the all-LDS selection kernel to the general one
    monkeypatch.setenv("SKH_TUNE_GREEDY_BIG_MIN", "4")                  # pairs with four or more candidate intervals select in global memory (greedy_big_kernel)
    monkeypatch.setenv("SKH_TUNE_SCAN_ONE_MAX", "16")                   # prefix sums over more than 16 values take the two-launch form (a ticketed reduce pass + the down pass) ...
    monkeypatch.setenv("SKH_TUNE_SCAN_TWO_MAX", "16")                   # ... here: the recursive form of arrays beyond 134 M values
    c = sk.Context(0)
    try:
        pc.case_triangle_synthetic(c, params=((1, 125),), length=60000)
    finally:
        c.close()
    monkeypatch.delenv("SKH_TUNE_SCAN_TWO_MAX")
    c = sk.Context(0)
    try:
        pc.case_triangle_synthetic(c, params=((1, 125), (0, 30)), length=60000)
        pc.case_screen_rules(c)
        pc.case_seeding_fixtures(c)
        pc.case_w_vs_w(c)
    finally:
        c.close()


def test_selection_in_global_memory(monkeypatch):
    """SKH_TUNE_GREEDY_BIG_MIN=2: every pair with two or more candidate intervals selects its chains with greedy_big_kernel (the kernel of pairs with
    thousands of candidates: its sort, its bin lists and its long-interval list in global memory)."""
    import importlib.util, os
    monkeypatch.setenv("SKH_TUNE_GREEDY_BIG_MIN", "2")
    c = sk.Context(0)
    try:
        pc.case_large_pair(c)
        pc.case_fragmented_genomes(c)
        pc.case_w_vs_w(c)
        pc.case_triangle_synthetic(c, params=((1, 125), (0, 30)), length=150000)
        spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
        fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
        rng = np.random.default_rng(99)
        assert sum(fz.one_round(c, rng, r) for r in range(20)) > 100
    finally:
        c.close()


def test_long_genome_pair(ctx):
    """A 26 Mbp pair at 3 % divergence: > 1024 chunks per pair, so finalize's per-pair work arrays leave LDS for the global scratch,
    and the pair alone fills several join super-batches' worth of tiles at a small budget."""
    from tests.helpers import random_genome, mutate, ora
    root = random_genome(26_000_000, 123)
    g = [[("a", root)], [("b", mutate(root, 0.03, 9))]]
    names = ["long0.fa", "long1.fa"]
    ss = ctx.sketch_records(g, sk.SketchParams(), names)
    osk = [ora.sketch_records(x, file_name=names[i]) for i, x in enumerate(g)]
    res, st = ctx.chain_pairs(ss, None, [0, 1], [1, 0], sk.MapParams(compute_ci=True), stats=True)
    for x, (i, j) in enumerate(((0, 1), (1, 0))):
        o, so = ora.chain_seeds(osk[i], osk[j], stats=True)
        pc.assert_result_close(res[x], o, (i, j))
        assert (int(st[x]["n_intervals"]), int(st[x]["n_accepted"]), int(st[x]["n_chunks"]), int(st[x]["n_estimates"]), int(st[x]["anchor_checksum"])) == \
            (so.n_intervals, so.n_accepted, so.n_chunks, so.n_estimates, so.anchor_checksum)
    assert int(st[0]["n_chunks"]) > 1024 and 0.96 < res[0]["ani"] < 0.98


def _wide_context(monkeypatch, span):
    monkeypatch.setenv("SKH_TUNE_WIDE_SPAN", str(span))
    return sk.Context(0)


def test_wide_sets_forced(monkeypatch):
    """SKH_TUNE_WIDE_SPAN=0 makes every sketch set a wide one -- the 64-bit coordinate path of genomes beyond 2^31 padded bases (include/skani_hip.h
    skh_sketch_is_wide): position indices in the tables and the join, widen_anchors_kernel, the 64-bit instantiations of the chunking, the sweep DP,
    the interval emission and the chunk statistics.  The parity cases must not notice."""
    import importlib.util, os
    c = _wide_context(monkeypatch, 0)
    try:
        probe = c.sketch_records([[("a", pc.random_genome(3000, 1))]], sk.SketchParams(), ["a.fa"])
        assert probe.wide
        pc.case_triangle_synthetic(c, params=((1, 125), (0, 30), (1, 8)), length=120000)
        pc.case_pinned_triples(c)
        pc.case_w_vs_w(c)
        pc.case_fragmented_genomes(c)
        pc.case_degenerate_pairs(c)
        pc.case_large_pair(c)
        pc.case_search_resident_db(c)
        pc.case_edge_cases_and_errors(c)
        spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
        fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
        rng = np.random.default_rng(77)
        assert sum(fz.one_round(c, rng, r) for r in range(15)) > 100
    finally:
        c.close()


def test_wide_set_meets_ordinary_sets(monkeypatch):
    """A threshold between the genome sizes: the set of the long genomes is wide, the set of the short ones is not, and pairs across the two sets
    (either one as the reference) take the mixed path -- one side's anchors carry position indices, the other's coordinates."""
    from tests.helpers import random_genome, mutate, ora
    c = _wide_context(monkeypatch, 300000)
    try:
        root = random_genome(400000, 5)
        longs = [[("l%d" % i, mutate(root, 0.02 + 0.01 * i, 40 + i))] for i in range(3)]
        shorts = [[("s%d" % i, mutate(root[50000 * i:50000 * i + 150000], 0.03, 50 + i)), ("t%d" % i, mutate(root[300000:340000], 0.01, 60 + i))] for i in range(3)]
        ln, sn = ["long%d.fa" % i for i in range(3)], ["short%d.fa" % i for i in range(3)]
        L = c.sketch_records(longs, sk.SketchParams(), ln); S = c.sketch_records(shorts, sk.SketchParams(), sn)
        assert L.wide and not S.wide
        ol = [ora.sketch_records(x, file_name=ln[i]) for i, x in enumerate(longs)]; os_ = [ora.sketch_records(x, file_name=sn[i]) for i, x in enumerate(shorts)]
        pr, pq = np.repeat(np.arange(3), 3), np.tile(np.arange(3), 3)
        for refs, queries, o_r, o_q in ((L, S, ol, os_), (S, L, os_, ol), (L, None, ol, ol)):
            res, st = c.chain_pairs(refs, queries, pr, pq, sk.MapParams(compute_ci=True), stats=True)
            for x in range(9):
                o, so = ora.chain_seeds(o_r[pr[x]], o_q[pq[x]], stats=True)
                pc.assert_result_close(res[x], o, (int(pr[x]), int(pq[x])))
                assert (int(st[x]["n_intervals"]), int(st[x]["n_accepted"]), int(st[x]["n_chunks"]), int(st[x]["anchor_checksum"])) == \
                    (so.n_intervals, so.n_accepted, so.n_chunks, so.anchor_checksum)
        # one set holding both kinds: the set is wide, its short genomes keep ordinary records, and a call over all its pairs splits into an
        # ordinary run (short against short) and a wide run (every pair with a long genome)
        M = c.sketch_records(longs + shorts, sk.SketchParams(), ln + sn); oall = ol + os_
        assert M.wide
        for g in range(6): pc.assert_sketch_equal(M, g, oall[g])
        pr = [i for i in range(6) for j in range(6)]; pq = [j for i in range(6) for j in range(6)]
        res, st = c.chain_pairs(M, None, pr, pq, sk.MapParams(compute_ci=True), stats=True)
        for x in range(36):
            o, so = ora.chain_seeds(oall[pr[x]], oall[pq[x]], stats=True)
            pc.assert_result_close(res[x], o, (pr[x], pq[x]))
            assert (int(st[x]["n_intervals"]), int(st[x]["n_accepted"]), int(st[x]["n_chunks"]), int(st[x]["anchor_checksum"])) == \
                (so.n_intervals, so.n_accepted, so.n_chunks, so.anchor_checksum)
        i, j, r, n = c.triangle(M, sk.MapParams())
        assert len(i) == 15
    finally:
        c.close()


def test_fragmented_genomes_beyond_31_bits(ctx):
    """No tunable here: two assemblies of 270,000 contigs of 600 bases are 162 Mbp of sequence but 2.37 G padded bases (8192 per contig), past the 31 bits
    of an ordinary set's coordinates.  Round 2 refused them; now the set is wide and the pair chains like any other."""
    from tests.helpers import big_random_genome, big_mutate, ora
    n_ctg, ln = 270000, 600
    a = big_random_genome(n_ctg * ln, 31); b = big_mutate(a, 0.02, 32)
    g = [[("c%d" % i, x[i * ln:(i + 1) * ln].tobytes()) for i in range(n_ctg)] for x in (a, b)]
    names = ["frag0.fa", "frag1.fa"]
    ss = ctx.sketch_records(g, sk.SketchParams(), names)
    assert ss.wide
    osk = [ora.sketch_records(x, file_name=names[i]) for i, x in enumerate(g)]
    pc.assert_sketch_equal(ss, 1, osk[1])
    res, st = ctx.chain_pairs(ss, None, [0, 1], [1, 0], sk.MapParams(compute_ci=True), stats=True)
    for x, (i, j) in enumerate(((0, 1), (1, 0))):
        o, so = ora.chain_seeds(osk[i], osk[j], stats=True)
        pc.assert_result_close(res[x], o, (i, j))
        assert (int(st[x]["n_intervals"]), int(st[x]["n_accepted"]), int(st[x]["n_chunks"]), int(st[x]["n_estimates"]), int(st[x]["anchor_checksum"])) == \
            (so.n_intervals, so.n_accepted, so.n_chunks, so.n_estimates, so.anchor_checksum)


def test_genome_pair_beyond_2_gbp(ctx):
    """A 2.3 Gbp genome in three contigs against its copy at 1 % divergence, and against a 3 Mbp piece of itself kept in an ordinary set: coordinates
    beyond 2^31 in the seeding, the tables, the join, the chaining and the statistics; 115,000 chunks in one pair."""
    from tests.helpers import big_random_genome, big_mutate, ora
    total = 2_300_000_000
    a = big_random_genome(total, 71); b = big_mutate(a, 0.01, 72)
    cuts = [0, 900_000_000, 1_700_000_000, total]
    g = [[("chr%d" % i, x[cuts[i]:cuts[i + 1]].tobytes()) for i in range(3)] for x in (a, b)]
    piece = [[("p", a[2_200_000_000:2_203_000_000].tobytes())]]
    del a, b
    names = ["giant0.fa", "giant1.fa"]
    import time
    t0 = time.time()
    ss = ctx.sketch_records(g, sk.SketchParams(), names)
    t_sketch = time.time() - t0
    small = ctx.sketch_records(piece, sk.SketchParams(), ["piece.fa"])
    assert ss.wide and not small.wide
    osk = [ora.sketch_records(x, file_name=names[i]) for i, x in enumerate(g)]; op = ora.sketch_records(piece[0], file_name="piece.fa")
    del g
    pc.assert_sketch_equal(ss, 1, osk[1])
    ctx.chain_pairs(ss, None, [0], [1], sk.MapParams(compute_ci=True))
    t0 = time.time()
    res = ctx.chain_pairs(ss, None, [0], [1], sk.MapParams(compute_ci=True))
    t_chain = time.time() - t0
    res, st = ctx.chain_pairs(ss, None, [0], [1], sk.MapParams(compute_ci=True), stats=True)
    t0 = time.time()
    o, so = ora.chain_seeds(osk[0], osk[1], stats=True)
    print("2.3 Gbp pair: pack + sketch (host buffers) %.2f s, chain %.3f s on the device; oracle chain_seeds %.2f s on one core" % (t_sketch, t_chain, time.time() - t0))
    pc.assert_result_close(res[0], o, (0, 1))
    assert (int(st[0]["n_intervals"]), int(st[0]["n_accepted"]), int(st[0]["n_chunks"]), int(st[0]["n_estimates"]), int(st[0]["anchor_checksum"])) == \
        (so.n_intervals, so.n_accepted, so.n_chunks, so.n_estimates, so.anchor_checksum)
    assert int(st[0]["n_chunks"]) > 100000 and 0.985 < res[0]["ani"] < 0.995
    for refs, queries, o_r, o_q in ((ss, small, osk[1], op), (small, ss, op, osk[1])):
        res, st = ctx.chain_pairs(refs, queries, [1 if refs is ss else 0], [0 if refs is ss else 1], sk.MapParams(compute_ci=True), stats=True)
        o, so = ora.chain_seeds(o_r, o_q, stats=True)
        pc.assert_result_close(res[0], o, (0, 0))
        assert (int(st[0]["n_accepted"]), int(st[0]["n_chunks"]), int(st[0]["anchor_checksum"])) == (so.n_accepted, so.n_chunks, so.anchor_checksum)


def test_randomised_differential(ctx):
    """tools/fuzz_parity.py: random genomes with duplications, inversions, N runs, many contigs; random c / k / m / seeding mode /
    estimator options; sketches, screens and every chaining stage against the oracle (600 rounds = 11,106 pairs were run clean
    when this was written; the suite runs 150: under a minute, mostly the oracle's time)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    rng = np.random.default_rng(2024)
    pairs = sum(fz.one_round(ctx, rng, r) for r in range(150))
    assert pairs > 1200


def test_two_contexts_share_a_sketch_set_across_threads(ctx):
    """Threading contract of the C ABI: a context is driven by one thread, distinct contexts are independent, sketch sets may be
    shared read-only.  Two host threads with their own contexts screen (lazy index cache) and chain the same set at the same time."""
    import threading
    genomes = pc.synthetic_clades(n_clades=3, members=4, length=150000, seed=91, tiny=False)
    names = ["t%02d.fa" % i for i in range(len(genomes))]
    ss = ctx.sketch_records(genomes, sk.SketchParams(), names)
    qs = ctx.sketch_records(genomes[:5], sk.SketchParams(), names[:5])
    mp = sk.MapParams(learned_ani=True, compute_ci=True)
    a0, b0 = ctx.screen(ss, qs, 0.0, 0, True)
    want = ctx.chain_pairs(ss, qs, b0, a0, mp)
    other = sk.Context(0)
    out, errs = {}, []
    def work(c, key):
        try:
            for _ in range(6):
                a, b = c.screen(ss, qs, 0.0, 0, True)
                r = c.chain_pairs(ss, qs, b, a, mp)
                assert np.array_equal(a, a0) and np.array_equal(b, b0) and r.tobytes() == want.tobytes()
            out[key] = True
        except Exception as e:                                 # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(ctx, "a")), threading.Thread(target=work, args=(other, "b"))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    other.close()
    assert not errs and out == {"a": True, "b": True}, errs


def test_screen_counter_planes_agree(monkeypatch):
    """The triangle screen keeps one copy of its count matrix per XCD and updates it with XCD-local atomics.  Near-threshold identities
    make the pass set sensitive to every single increment: it must equal the one from a single device-scope matrix."""
    from tests.parity_cases import synthetic_clades
    genomes = synthetic_clades(n_clades=6, members=16, length=150000, seed=97, tiny=False)
    got = {}
    for planes in ("8", "1"):
        monkeypatch.setenv("SKH_TUNE_SCREEN_PLANES", planes)
        c = sk.Context(0)
        try:
            ss = c.sketch_records(genomes, sk.SketchParams(marker_c=200), None)    # ~750 markers per genome: dense rows
            got[planes] = [c.screen(ss, None, identity=x) for x in (0.80, 0.93, 0.96, 0.98, 0.995)]
        finally:
            c.close()
    for a, b in zip(got["8"], got["1"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    sizes = [len(a[0]) for a in got["8"]]
    assert sizes[0] > 0 and sizes[0] > sizes[-1] and len(set(sizes)) >= 3, sizes      # the identities do separate the pairs



def test_sketch_offsets_exact_while_the_gpu_is_busy():
    """The per-genome offsets a sketch call reads back must not depend on what else the GPU is doing: the same genomes are sketched 25 times while three other
    streams of this process keep the device busy with short kernels, and every run's offsets equal the quiet run's.
    What this test is NOT: a reproducer of round 4's missing barrier in seed_offsets_kernel (HISTORY.md section 5b-19).  The library without that barrier passes it too
    (checked on an MI355X) -- only eight PROCESSES sharing the GPU brought that race out (tests/test_zz_bench_multirank.py, test_config4_eight_ranks_one_device)."""
    import threading
    import torch
    import bench
    dev = torch.device("cuda", 0)
    c = sk.Context(0)
    try:
        bases, coff, cgen, ng, _ = bench.make_genomes(torch, dev, np.arange(300), mean_len=1_000_000)
        torch.cuda.synchronize()
        gs = c.pack_buffer(None, coff, cgen, ng, sk.SEED_AVX2, device_ptr=bases.data_ptr())
        del bases
        params = sk.SketchParams()
        def offsets():
            ss = c.sketch_genomes(gs, params, genome_rank=np.arange(ng, dtype=np.uint32))
            try:
                m = ss.export_meta()
                return m["pos_off"].copy(), m["marker_off"].copy()
            finally:
                ss.close()
        quiet = offsets()
        stop = threading.Event()
        def noise():
            s = torch.cuda.Stream(device=dev)
            a = torch.ones(1 << 16, device=dev)
            with torch.cuda.stream(s):
                while not stop.is_set():
                    for _ in range(200):
                        a = a * 1.0001 + 1.0
                    s.synchronize()
        th = [threading.Thread(target=noise) for _ in range(3)]
        for t in th:
            t.start()
        try:
            for rep in range(25):
                busy = offsets()
                assert np.array_equal(busy[0], quiet[0]) and np.array_equal(busy[1], quiet[1]), rep
        finally:
            stop.set()
            for t in th:
                t.join()
        gs.close()
    finally:
        c.close()
        torch.cuda.empty_cache()
