"""Parity cases shared by the GPU tests (tests/test_gpu_parity.py, through libskani_hip.so on an MI355X) and by the
kernel-simulator tests (tests/test_emu_pipeline.py, same kernel sources under tests/emu on the CPU).
Every case compares the C-ABI path with the CPU oracle on identical inputs.  Integer outputs must be bit-exact;
ANI/AF are compared with the north-star tolerance 1e-4 (they are in fact expected to be identical to f32 rounding)."""
import numpy as np
import pytest

import skani_amd as sk
from tests.helpers import (MODEL_C125, MODEL_C200, golden_records, mutate, o157_arrays, ora, oracle_o157, oracle_sketch_file,
                           pinned, random_genome)

TOL = 1e-4
FLOAT_FIELDS = ("ani", "af_query", "af_ref", "ci_lower", "ci_upper", "std")
EXACT_FIELDS = ("q90_q", "q90_r", "q50_q", "q50_r", "q10_q", "q10_r", "num_contigs_q", "num_contigs_r", "avg_chain_int_len",
                "total_bases_covered")


def assert_sketch_equal(ss, g, osk):
    e = ss.export(g)
    s, p, cc = osk.seeds(pos_order=True)
    assert np.array_equal(e["seed"], s) and np.array_equal(e["pos"], p) and np.array_equal(e["ctgcanon"], cc), "seed records differ"
    assert np.array_equal(e["markers"], osk.markers()), "marker set differs"
    assert np.array_equal(e["contig_lengths"], osk.contig_lengths())
    sz = ss.sizes(g)
    assert (sz["n_pos"], sz["n_distinct"], sz["n_markers"], sz["total_len"]) == (osk.n_positions, osk.n_distinct, osk.n_markers, osk.total_len)


def assert_result_close(got, want, ctx=""):
    """got: numpy record (skh_ani_result); want: oracle AniResult or numpy record."""
    for f in FLOAT_FIELDS:
        a = float(got[f]); b = float(want[f]) if isinstance(want, np.void) else float(getattr(want, f))
        if np.isnan(b):
            assert np.isnan(a), (ctx, f, a, b)
        else:
            assert abs(a - b) <= TOL, (ctx, f, a, b)
    for f in EXACT_FIELDS:
        a = got[f]; b = want[f] if isinstance(want, np.void) else getattr(want, f)
        assert a == b, (ctx, f, a, b)


def import_o157(ctx, params=None):
    z = o157_arrays()
    return ctx.import_sketches(params or sk.SketchParams(), [dict(seed=z["seed"], pos=z["pos"], ctgcanon=z["ctgcanon"], markers=z["markers"],
                                                                  contig_lengths=z["contig_lengths"], total_len=int(z["total_len"]))],
                               names=[str(z["file_name"])], genome_rank=[1000])


# ---------------------------------------------------------------------------------------------------- seeding
def case_seeding_golden_plasmid(ctx):
    """759 seed records + 81 markers of the reference's golden sketch (both seeding semantics agree here)."""
    z = o157_arrays(); sel = (z["ctgcanon"] >> 1) == 1
    o = np.lexsort((z["pos"][sel],))
    for mode in (sk.SEED_SCALAR, sk.SEED_AVX2):
        ss = ctx.sketch_records([golden_records("o157_plasmid.fasta")], sk.SketchParams(seeding_mode=mode), ["p"])
        e = ss.export(0)
        assert np.array_equal(e["seed"], z["seed"][sel][o]) and np.array_equal(e["pos"], z["pos"][sel][o])
        assert np.array_equal(e["ctgcanon"] & 1, z["ctgcanon"][sel][o] & 1)
        assert len(e["markers"]) == 81 and np.isin(e["markers"], z["markers"]).all()


def case_seeding_fixtures(ctx):
    """viruses.fna ((L-20)%4 != 0: AVX2 tail rule), all-N, tiny/short contigs, N and n runs, k=16, c=30/m=200."""
    V = golden_records("viruses.fna"); P = golden_records("o157_plasmid.fasta"); N = golden_records("all_ns.fa")
    mixed = [("a", random_genome(30011, 1, 0.001)), ("short", random_genome(777, 2)), ("c", random_genome(50003, 3, 0.01)),
             ("tiny", random_genome(120, 4))]
    lower = [("x", random_genome(20000, 5, 0.002).replace(b"N", b"n", 10))]
    for mode in (sk.SEED_SCALAR, sk.SEED_AVX2):
        for (c, k, m) in ((125, 15, 1000), (30, 16, 200), (70, 15, 1000), (200, 14, 1000)):
            genomes = [V, P, mixed, lower, N]
            ss = ctx.sketch_records(genomes, sk.SketchParams(c, k, m, mode), [str(i) for i in range(len(genomes))])
            assert len(ss) == len(genomes)
            for g, recs in enumerate(genomes):
                assert_sketch_equal(ss, g, ora.sketch_records(recs, c, k, m, str(g), mode))
    ss = ctx.sketch_records([N], sk.SketchParams(), ["n"])
    assert ss.sizes(0)["n_pos"] == 0


def case_pack_every_byte(ctx):
    """The ingest kernel's byte handling (types.rs:40-49 BYTE_TO_SEQ, N detection of both seeding paths): contigs drawn from ACGT with lower case,
    N / n runs, U / u, IUPAC letters, punctuation, the raw codes 0..3 and high bytes, with lengths that put the following contigs at every byte
    alignment; dense sketches (c = 6) so that nearly every window is looked at; also through a device buffer at an odd address."""
    rng = np.random.default_rng(31)
    common = np.frombuffer(b"ACGT", np.uint8)
    odd = np.frombuffer(b"acgtNnUuRYKMSWBDHVryx-*. \x00\x01\x02\x03\x7f\xff@[`{", np.uint8)
    recs = []
    for i in range(48):
        L = int(rng.integers(520, 4000)) + i % 7
        a = common[rng.integers(0, 4, L)]
        where = rng.random(L) < (0.0 if i % 5 == 0 else 0.03)
        a = np.where(where, odd[rng.integers(0, len(odd), L)], a).astype(np.uint8)
        if i % 4 == 1:
            st = int(rng.integers(0, L - 100)); a[st:st + 40] = ord("N")
        if i % 4 == 2:
            st = int(rng.integers(0, L - 100)); a[st:st + 33] = ord("n")
        recs.append(("c%d" % i, a.tobytes()))
    genomes = [recs[:20], recs[20:33], recs[33:]]
    for mode in (sk.SEED_SCALAR, sk.SEED_AVX2):
        ss = ctx.sketch_records(genomes, sk.SketchParams(6, 15, 24, mode), ["p%d" % g for g in range(3)])
        for g, r in enumerate(genomes):
            assert_sketch_equal(ss, g, ora.sketch_records(r, 6, 15, 24, "p%d" % g, mode))


def case_pack_in_batches(ctx):
    """skh_genomes_begin / _append / _finish (the streaming ingest of `skani-hip triangle`): the same genomes packed in one call and in three batches from
    pinned buffers with gaps between the contigs, the genomes arriving out of order -- one of them without contigs, one a tiny contig -- give the same
    sketches; the capacity announced at the beginning and the one-batch-per-genome rule are enforced."""
    genomes = synthetic_clades(n_clades=2, members=3, length=40000, seed=23, tiny=True)
    kept = [[s for _, s in g if len(s) >= 500] for g in genomes] + [[]]
    whole = ctx.pack_genomes(kept, sk.SEED_AVX2)
    numbered = list(enumerate(kept))
    parts = ctx.pack_batches([numbered[5:], numbered[:2], numbered[2:5][::-1]], sk.SEED_AVX2, max_bases=sum(len(s) for g in kept for s in g) + 1000, max_contigs=64)   # any order
    assert whole.total_bases == parts.total_bases
    a = ctx.sketch_genomes(whole, sk.SketchParams(c=30, marker_c=200)); b = ctx.sketch_genomes(parts, sk.SketchParams(c=30, marker_c=200))
    assert len(a) == len(b) == len(kept)
    for g in range(len(kept)):
        ea, eb = a.export(g), b.export(g)
        for key in ("seed", "pos", "ctgcanon", "markers", "contig_lengths"):
            assert np.array_equal(ea[key], eb[key]), (g, key)
    osk = ora.sketch_records(genomes[1], 30, 15, 200, "g1", 1)
    assert_sketch_equal(b, 1, osk)
    with pytest.raises(sk.SkaniHipError):
        ctx.pack_batches([numbered[:2], numbered[2:]], sk.SEED_AVX2, max_bases=1000, max_contigs=64)
    with pytest.raises(sk.SkaniHipError):                                       # a genome's contigs in two batches
        ctx.pack_batches([[(0, kept[0][:1])], [(0, kept[0][1:] or kept[1])]], sk.SEED_AVX2)


def case_seeding_ecoli_w(ctx):
    W = golden_records("e.coli-W.fasta.gz")
    ss = ctx.sketch_records([W], sk.SketchParams(), ["w"])
    osk = ora.sketch_records(W, mode=1)
    assert_sketch_equal(ss, 0, osk)
    sz = ss.sizes(0)
    assert (sz["n_pos"], sz["n_distinct"], sz["n_markers"]) == (39310, 37786, 4649)


def case_seeding_low_complexity(ctx):
    """A period-2/3/7 repeat makes every window of a region hit => exercises worst-case tile capacity and table multiplicities."""
    rng = np.random.default_rng(5)
    found = None
    thr = (2**64 - 1) // 30
    for _ in range(4000):
        unit = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 7))
        rep = unit * 40
        sk_o = ora.Sketch(30, 15, 1000, ""); sk_o.add_contig(rep, 0, 0)
        if sk_o.n_positions > 30:
            found = unit; break
    assert found is not None
    seq = random_genome(5000, 1) + found * 3000 + random_genome(5000, 2)
    for mode in (0, 1):
        ss = ctx.sketch_records([[("lc", seq)]], sk.SketchParams(30, 15, 1000, mode), ["lc"])
        assert_sketch_equal(ss, 0, ora.sketch_records([("lc", seq)], 30, 15, 1000, "lc", mode))
        assert ss.sizes(0)["n_pos"] > 2000


# ---------------------------------------------------------------------------------------------------- chain
def case_pinned_triples(ctx):
    """reference test_results_versions/0.3.0:130-135 (search --median; learned ANI off) through the GPU path."""
    names = ["test_files/e.coli-W.fasta", "test_files/o157_plasmid.fasta"]
    refs = ctx.sketch_records([golden_records("e.coli-W.fasta.gz"), golden_records("o157_plasmid.fasta")], sk.SketchParams(), names)
    o157 = import_o157(ctx)
    mp = sk.MapParams(median=True, compute_ci=True)
    res, st = ctx.chain_pairs(refs, o157, [1, 0], [0, 0], mp, stats=True)
    self_res = ctx.chain_pairs(o157, None, [0], [0], mp)
    got = [("%.2f" % (r["ani"] * 100), "%.2f" % (r["af_ref"] * 100), "%.2f" % (r["af_query"] * 100)) for r in list(res) + list(self_res)]
    want = [("%.2f" % t["ani"], "%.2f" % t["af_ref"], "%.2f" % t["af_query"]) for t in pinned()["triples_median_percent"]]
    assert got == want, (got, want)
    assert (int(st[1]["n_anchors"]), int(st[1]["n_chunks"]), int(st[1]["n_intervals"]), int(st[1]["n_accepted"]), int(st[1]["n_estimates"])) == \
        (31989, 245, 500, 380, 234)
    # and against the oracle, field by field, incl. the stage checksums
    ow = oracle_sketch_file("e.coli-W.fasta.gz", file_name=names[0]); op = oracle_sketch_file("o157_plasmid.fasta", file_name=names[1]); oo = oracle_o157()
    for i, oref in ((0, op), (1, ow)):
        r, s = ora.chain_seeds(oref, oo, median=True, stats=True)
        assert_result_close(res[i], r, i)
        assert (int(st[i]["switched"]), int(st[i]["n_anchors"]), int(st[i]["n_qpos"]), int(st[i]["anchor_checksum"])) == \
            (s.switched, s.n_anchors, s.n_qpos, s.anchor_checksum)


def case_w_vs_w(ctx):
    """tests/tests.rs:42-60"""
    W = golden_records("e.coli-W.fasta.gz")
    ss = ctx.sketch_records([W], sk.SketchParams(), ["w"])
    r = ctx.chain_pairs(ss, None, [0], [0], sk.MapParams())[0]
    assert r["ani"] >= 1.0 and r["af_query"] >= 0.99 and r["af_ref"] >= 0.99


def case_viruses_individual(ctx):
    """tests/int_test_new.rs:57-62 (triangle -i): one sketch per contig."""
    recs = golden_records("viruses.fna")
    ss = ctx.sketch_records([[r] for r in recs], sk.SketchParams(), ["test_files/viruses.fna"] * len(recs))
    osk = []
    for name, seq in recs:
        s = ora.Sketch(125, 15, 1000, "test_files/viruses.fna"); s.add_contig(seq, 1); osk.append(s)
    pr = [0, 0, 1]; pq = [1, 2, 2]
    res = ctx.chain_pairs(ss, None, pr, pq, sk.MapParams(compute_ci=True))
    anis = []
    for x, (i, j) in enumerate(zip(pr, pq)):
        assert_result_close(res[x], ora.chain_seeds(osk[i], osk[j]), (i, j))
        if res[x]["ani"] > 0.1:
            anis.append(res[x]["ani"] * 100)
    assert any(99.0 < a < 99.9 for a in anis) and any(a > 99.9 for a in anis), anis


def synthetic_clades(n_clades=2, members=3, length=200000, seed=11, tiny=True):
    genomes = []
    for cl in range(n_clades):
        root = random_genome(length, seed + cl)
        for m in range(members):
            s = mutate(root, 0.005 + 0.02 * m, (seed + cl) * 100 + m)
            cuts = sorted(np.random.default_rng((seed + cl) * 7 + m).integers(1000, length - 1000, m))
            ctgs, prev = [], 0
            for c in list(cuts) + [length]:
                ctgs.append(("c%d" % len(ctgs), s[prev:c])); prev = c
            genomes.append(ctgs)
    if tiny:
        genomes.append([("tiny", random_genome(3000, 99))])
    return genomes


def key_range_fits(ctx, ss):
    """False when the context's screen budget (SKH_TUNE_SCREEN_CELLS in the small-budget tests) is below the set's dense count matrix: skh_screen_part then refuses."""
    try:
        ctx.screen_part(ss, 0, 1); return True
    except sk.SkaniHipError as e:
        assert "beyond the screen's budget" in str(e); return False


def case_triangle_synthetic(ctx, params=((1, 125), (0, 30), (1, 70), (1, 200), (0, 20)), length=200000):
    """triangle.rs:55-105 on synthetic clades: screen pass set, chained pairs, all result fields, learned ANI on/off,
    robust/median windows.  c = 20 (band 125) goes through the wave-sweep chaining kernels, the others through the thread-per-chunk one."""
    genomes = synthetic_clades(length=length)
    names = ["g%02d.fa" % i for i in range(len(genomes))]
    for mode, c in params:
        ss = ctx.sketch_records(genomes, sk.SketchParams(c=c, seeding_mode=mode), names)
        osk = [ora.sketch_records(g, c, 15, 1000, names[i], mode) for i, g in enumerate(genomes)]
        a, b = ctx.screen(ss, None, 0.0, 0, True)
        exp = [(i, int(j)) for i in range(len(osk) - 1) for j in ora.screen_refs(osk, osk[i], 0.8, 0, True) if j > i]
        assert list(zip(a.tolist(), b.tolist())) == sorted(exp)
        for n_parts in (1, 3, 8) if key_range_fits(ctx, ss) else ():                 # the same screen cut by key range: the parts' cells add up to the full count matrix
            cells = np.concatenate([ctx.screen_part(ss, part, n_parts) for part in range(n_parts)])
            a3, b3 = ctx.screen_from_cells(ss, cells, 0.0, True)
            ci, cj, cc = ctx.unpack_cells(cells)
            assert list(zip(a3.tolist(), b3.tolist())) == sorted(exp) and (ci < cj).all() and (cc > 0).all(), (mode, c, n_parts)
            # cells in row order are added up row by row in LDS (what the distributed triangle's gathered blocks get); the concatenation of several parts is not in row
            # order and takes the dense matrix: the same list either way (a cell of a pair appears once per part that saw one of its markers)
            a4, b4 = ctx.screen_from_cells(ss, np.sort(cells), 0.0, True)
            assert np.array_equal(a4, a3) and np.array_equal(b4, b3), (mode, c, n_parts)
            if n_parts == 3:                                                          # a cell that names a genome beyond the set is refused, not added somewhere
                bad = np.append(cells, np.uint64((len(osk) << 43) | (1 << 22) | 5))
                try:
                    ctx.screen_from_cells(ss, bad, 0.0, True); raise AssertionError("cell beyond the set accepted")
                except sk.SkaniHipError as e:
                    assert "beyond the set" in str(e)
        for r0, nr in ((0, len(osk)), (1, 3), (len(osk) - 1, 1), (2, 0)):             # row blocks of the same screen
            a2, b2 = ctx.screen_rows(ss, r0, nr, 0.0, True)
            assert list(zip(a2.tolist(), b2.tolist())) == [e for e in sorted(exp) if r0 <= e[0] < r0 + nr]
        learned = sk.use_learned_ani(c)
        model = ora.Model(MODEL_C125 if abs(c - 125) < abs(c - 200) else MODEL_C200) if learned else None
        for kw in (dict(), dict(robust=True), dict(median=True)):
            use_model = learned and not kw.get("median")
            mp = sk.MapParams(learned_ani=use_model, compute_ci=True, **kw)
            i, j, res, nch = ctx.triangle(ss, mp)
            oi, oj, ores, onch, _ = ora.triangle(osk, model=model if use_model else None, **kw)
            assert nch == onch and np.array_equal(i, oi) and np.array_equal(j, oj), (mode, c, kw)
            for x in range(len(res)):
                assert_result_close(res[x], ores[x], (mode, c, kw, int(i[x]), int(j[x])))
            if not kw:
                # sets sketched with deferred seed tables: the triangle builds the tables beside its screen (skh_triangle); with and without the screen
                # index made at sketch time; a second triangle on the same set finds the tables built.  Byte for byte the result above.
                # compact: SKH_SKETCH_COMPACT | SKH_SKETCH_DEFER_TABLES -- the build then also MOVES the list storage, which the pair descriptors made beside it point into
                for screen_index, compact in ((True, False), (False, False), (True, True)):
                    ssd = ctx.sketch_records(genomes, sk.SketchParams(c=c, seeding_mode=mode), names, defer_tables=True, screen_index=screen_index, compact=compact)
                    for again in range(2):
                        di, dj, dres, dn = ctx.triangle(ssd, mp)
                        assert dn == nch and np.array_equal(di, i) and np.array_equal(dj, j) and dres.tobytes() == res.tobytes(), (mode, c, screen_index, compact, again)
                    ssd.close()


def case_screen_rules(ctx):
    """screen.rs rules 0/1/2, query-vs-ref and triangle forms, rescue_small on/off."""
    genomes = synthetic_clades(n_clades=2, members=2, length=60000, seed=31) + [[("few", random_genome(9000, 77))]]
    names = ["s%02d.fa" % i for i in range(len(genomes))]
    queries = [genomes[0], [("q", mutate(genomes[2][0][1], 0.03, 5))], [("none", random_genome(40000, 1234))]]
    for m in (1000, 200):
        sp = sk.SketchParams(marker_c=m)
        refs = ctx.sketch_records(genomes, sp, names); qs = ctx.sketch_records(queries, sp, ["q0", "q1", "q2"])
        orefs = [ora.sketch_records(g, 125, 15, m, names[i], 1) for i, g in enumerate(genomes)]
        oqs = [ora.sketch_records(g, 125, 15, m, "q%d" % i, 1) for i, g in enumerate(queries)]
        for rescue in (True, False):
            for rule in (0, 2):
                a, b = ctx.screen(refs, qs, 0.8, rule, rescue)
                exp = [(q, int(r)) for q in range(len(oqs)) for r in ora.screen_refs(orefs, oqs[q], 0.8, rule, rescue)]
                assert list(zip(a.tolist(), b.tolist())) == sorted(exp), (m, rescue, rule)
            a, b = ctx.screen(refs, qs, 0.8, 1, rescue)
            exp = [(q, r) for q in range(len(oqs)) for r in range(len(orefs)) if ora.check_markers_quickly(orefs[r], oqs[q], 0.8, rescue)]
            assert list(zip(a.tolist(), b.tolist())) == sorted(exp), (m, rescue, "quick")
            a, b = ctx.screen(refs, None, 0.8, 0, rescue)
            exp = [(i, int(j)) for i in range(len(orefs) - 1) for j in ora.screen_refs(orefs, orefs[i], 0.8, 0, rescue) if j > i]
            assert list(zip(a.tolist(), b.tolist())) == sorted(exp), (m, rescue, "tri")
            if not key_range_fits(ctx, refs): continue
            cells = np.concatenate([ctx.screen_part(refs, part, 4) for part in range(4)])   # the triangle cut by key range (the small genome's rescue: a row that passes without a count)
            for form in (cells, np.sort(cells)):                                      # as concatenated (dense matrix) / in row order (row by row in LDS)
                a, b = ctx.screen_from_cells(refs, form, 0.8, rescue)
                assert list(zip(a.tolist(), b.tolist())) == sorted(exp), (m, rescue, "tri by key range")


def case_screen_marker_prefix_groups(ctx):
    """The screen sorts its (marker, genome) incidences by the marker's leading 32 bits only: markers that differ in their low 10 bits share a prefix
    group and must not be counted as shared.  Imported sketches with crafted marker sets: g0 and g1 share 600 markers exactly; g2 holds the same 600
    prefixes with OTHER low bits (shares nothing); g3 shares 300 of g0's markers and has 300 same-prefix look-alikes."""
    rng = np.random.default_rng(5)
    pre = np.unique(rng.integers(0, 1 << 32, 600, dtype=np.uint64))[:600]
    low = rng.integers(0, 512, len(pre), dtype=np.uint64)
    m0 = (pre << np.uint64(10)) | low
    sets = [m0, m0.copy(), (pre << np.uint64(10)) | (low + np.uint64(1)), np.concatenate([m0[:300], ((pre[300:] << np.uint64(10)) | (low[300:] ^ np.uint64(512)))])]
    recs, osk = [], []
    for g, mk in enumerate(sets):
        n = 50
        rec = dict(seed=rng.integers(0, 1 << 30, n, dtype=np.uint32), pos=np.arange(n, dtype=np.uint32) * 100, ctgcanon=np.zeros(n, np.uint32), markers=np.sort(mk),
                   contig_lengths=np.array([600000], np.uint32), total_len=600000)
        recs.append(rec)
        osk.append(ora.Sketch.from_arrays(125, 15, 1000, "g%d" % g, rec["seed"], rec["pos"], rec["ctgcanon"], rec["markers"], rec["contig_lengths"], rec["total_len"]))
    names = ["g%d" % g for g in range(len(sets))]
    refs = ctx.import_sketches(sk.SketchParams(), recs, names=names)
    for rescue in (True, False):
        a, b = ctx.screen(refs, None, 0.8, 0, rescue)
        exp = [(i, int(j)) for i in range(len(osk) - 1) for j in ora.screen_refs(osk, osk[i], 0.8, 0, rescue) if j > i]
        assert list(zip(a.tolist(), b.tolist())) == sorted(exp), ("tri", rescue, list(zip(a.tolist(), b.tolist())), sorted(exp))
        assert (0, 1) in exp and (0, 2) not in exp and (1, 2) not in exp
        for rule in (0, 2):
            a, b = ctx.screen(refs, refs, 0.8, rule, rescue)
            exp = [(q, int(r)) for q in range(len(osk)) for r in ora.screen_refs(osk, osk[q], 0.8, rule, rescue)]
            assert list(zip(a.tolist(), b.tolist())) == sorted(exp), ("qr", rule, rescue)
    refs.close()


def _marker_set_import(ctx, sets):
    rng = np.random.default_rng(11)
    recs = []
    for mk in sets:
        n = 20
        recs.append(dict(seed=rng.integers(0, 1 << 30, n, dtype=np.uint32), pos=np.arange(n, dtype=np.uint32) * 100, ctgcanon=np.zeros(n, np.uint32), markers=np.sort(mk),
                         contig_lengths=np.array([600000], np.uint32), total_len=600000))
    return ctx.import_sketches(sk.SketchParams(), recs, names=["g%04d" % g for g in range(len(sets))])


def _shared_marker_counts(sets):
    """(i, j) -> number of common markers, i < j, exactly (sparse incidence matrix times its transpose)."""
    import scipy.sparse as sp
    allm, inv = np.unique(np.concatenate(sets), return_inverse=True)
    rows = np.concatenate([np.full(len(m), g, np.int64) for g, m in enumerate(sets)])
    A = sp.csr_matrix((np.ones(len(rows), np.int64), (rows, inv)), shape=(len(sets), len(allm)))
    C = sp.triu(A @ A.T, k=1).tocoo()
    return {(int(i), int(j)): int(c) for i, j, c in zip(C.row, C.col, C.data) if c}


def case_screen_incidence_sort(make_ctx):
    """The screen's incidence list is sorted by a hand-written bucket sort (screen_keys.hip): tiles of (bucket range) x (genome group) count and scatter the keys,
    a workgroup per bucket orders them.  Crafted marker sets -- canonical-k-mer-like values (density 2(1 - x)), clades sharing most of their markers, one marker in
    every genome (a fine group of one prefix), two neighbouring prefixes in a hundred genomes each (a long fine group that has to be ranked), same-prefix look-alikes
    -- through the key-range screen, whose cells are the count matrix itself: every count must equal the exact number of common markers.  Run with the default
    bucket size, with tiny buckets (several bucket ranges per genome, thousands of buckets), with an LDS capacity the largest bucket exceeds (the radix-sort way
    out), with the radix sort alone (the form of rounds 1-4), and with two large genomes (bucket ranges narrower than the bucket count: one genome per tile)."""
    rng = np.random.default_rng(2026)
    def canon(n): return np.minimum(rng.integers(0, 1 << 42, n, dtype=np.uint64), rng.integers(0, 1 << 42, n, dtype=np.uint64))
    G, CL = 160, 8
    universal = np.uint64(0x2AAAAAAAAAA)
    twin = (np.uint64(123456789) << np.uint64(10)) | np.uint64(5)
    twins = [twin, twin + np.uint64(1 << 10)]                                          # neighbouring prefixes
    sets = []
    for c in range(G // CL):
        base = canon(300)
        for m in range(CL):
            g = c * CL + m
            own = [base[rng.random(len(base)) < 0.7], canon(60), [universal]]
            if g < 100: own.append([twins[0]])
            if 60 <= g: own.append([twins[1]])
            if g % 5 == 0: own.append(base[:20] ^ np.uint64(1))                          # look-alikes: same prefix, other low bits
            sets.append(np.unique(np.concatenate([np.asarray(x, np.uint64) for x in own])))
    big = [np.unique(canon(150000)) for _ in range(2)]
    big[1] = np.unique(np.concatenate([big[1], big[0][::3]]))
    runs = (({}, sets), ({"SKH_TUNE_SKEYS_AVG": "16"}, sets), ({"SKH_TUNE_SKEYS_AVG": "16", "SKH_TUNE_SKEYS_CAP": "64"}, sets), ({"SKH_TUNE_SCREEN_SORT_RADIX": "1"}, sets),
            ({}, big), ({"SKH_TUNE_SKEYS_AVG": "100"}, big))
    pairs_seen = {}
    for env, ms in runs:
        want = _shared_marker_counts(ms)
        ctx = make_ctx(env)
        try:
            refs = _marker_set_import(ctx, ms)
            for n_parts in (1, 3):
                cells = np.concatenate([ctx.screen_part(refs, part, n_parts) for part in range(n_parts)])
                i, j, c = ctx.unpack_cells(cells)
                got = {}
                for a, b, n in zip(i.tolist(), j.tolist(), c.tolist()): got[(a, b)] = got.get((a, b), 0) + n
                assert got == want, (env, n_parts, len(got), len(want))
            tri = tuple(map(tuple, (x.tolist() for x in ctx.screen(refs, None, 0.8, 0, True))))
            qr = tuple(map(tuple, (x.tolist() for x in ctx.screen(refs, refs, 0.8, 0, True))))
            key = len(ms)
            if key in pairs_seen: assert pairs_seen[key] == (tri, qr), env
            pairs_seen[key] = (tri, qr)
            assert len(tri[0]) > 0
            refs.close()
        finally:
            ctx.close()


def case_screen_count_walks(make_ctx, G=1700):
    """The triangle's count kernel stages 1024 keys and 256 on either side in LDS; the lanes of a marker's group walk the group together from its first incidence
    (screen.hip screen_count_tri_rows_kernel), and what lies beyond the staged keys is read from global memory.  Crafted sets: one marker in ALL genomes (a group several
    tiles long, reached from both sides), one in 600 of them, clades of eight sharing most of their markers, same-prefix look-alikes.  Every cell of the key-range
    screen must hold the exact number of common markers, with this kernel and with the one of rounds 1-5 (SKH_TUNE_SCREEN_COUNT_ROWS=0), and both must give
    the same candidate list."""
    rng = np.random.default_rng(606)
    def canon(n): return np.minimum(rng.integers(0, 1 << 42, n, dtype=np.uint64), rng.integers(0, 1 << 42, n, dtype=np.uint64))
    universal, wide = np.uint64(0x155555555AA), np.uint64((987654321 << 10) | 77)
    sets = []
    for c in range(G // 8):
        base = canon(40)
        for m in range(8):
            g = c * 8 + m
            own = [base[rng.random(len(base)) < 0.8], canon(6), [universal]]
            if 300 <= g < 900: own.append([wide])
            if g % 7 == 0: own.append(base[:5] ^ np.uint64(3))
            sets.append(np.unique(np.concatenate([np.asarray(x, np.uint64) for x in own])))
    sets = [sets[g] for g in rng.permutation(len(sets))]                               # related genomes far apart: the count matrix's column order brings them together
    want = _shared_marker_counts(sets)
    seen = []
    # (COL_ORDER=2: the column order also inside the key-range parts, which make it only from ~1,000 incidences per genome on)
    for env in ({}, {"SKH_TUNE_SCREEN_COUNT_ROWS": "0"}, {"SKH_TUNE_SCREEN_PLANES": "1", "SKH_TUNE_SCREEN_COL_ORDER": "2"}, {"SKH_TUNE_SCREEN_COL_ORDER": "0"}, {"SKH_TUNE_SCREEN_COL_ORDER": "2"}):
        ctx = make_ctx(env)
        try:
            refs = _marker_set_import(ctx, sets)
            for n_parts in (1, 3):
                cells = np.concatenate([ctx.screen_part(refs, part, n_parts) for part in range(n_parts)])
                i, j, c = ctx.unpack_cells(cells)
                got = {}
                for a, b, n in zip(i.tolist(), j.tolist(), c.tolist()): got[(a, b)] = got.get((a, b), 0) + n
                assert got == want, (env, n_parts, len(got), len(want))
            seen.append(tuple(x.tobytes() for x in ctx.screen(refs, None, 0.8, 0, False)) + tuple(x.tobytes() for x in ctx.screen_rows(refs, 5, G // 2, 0.8, False)))
            refs.close()
        finally:
            ctx.close()
    assert all(x == seen[0] for x in seen) and len(seen[0][0]) > 0


def case_screen_from_cells_large_rows(ctx, N=17000):
    """The gathered cells of the key-range screen are added up row by row in an LDS row of N counters (screen_rows_from_cells_kernel): beyond 16,384 genomes that row is
    more than the 64 KB a launch may ask for unannounced.  17,000 genomes with a dozen markers each, clades of four: the candidate list from the cells equals the
    list of the one-call screen and the exact one."""
    rng = np.random.default_rng(17)
    sets = []
    for c in range(N // 4):
        base = np.minimum(rng.integers(0, 1 << 42, 9, dtype=np.uint64), rng.integers(0, 1 << 42, 9, dtype=np.uint64))
        for m in range(4):
            sets.append(np.unique(np.concatenate([base[: 9 - m], rng.integers(0, 1 << 42, 3, dtype=np.uint64)])))
    meta = dict(pos_off=np.arange(N + 1, dtype=np.uint64), marker_off=np.concatenate([[0], np.cumsum([len(x) for x in sets])]).astype(np.uint64),
                contig_off=np.arange(N + 1, dtype=np.uint64), contig_lengths=np.full(N, 600000, np.uint32), total_len=np.full(N, 600000, np.uint64),
                genome_rank=np.arange(N, dtype=np.uint32))
    refs = ctx.import_flat(sk.SketchParams(), meta, seed=rng.integers(0, 1 << 30, N, dtype=np.uint32), pos=np.full(N, 100, np.uint32), ctgcanon=np.zeros(N, np.uint32),
                           markers=np.concatenate(sets))
    try:
        want = sorted(p for p, n in _shared_marker_counts(sets).items() if n > 1)    # screen.rs:176-187: more than max(floor(0.8^21 x 12), 1) common markers
        a, b = ctx.screen(refs, None, 0.8, 0, False)
        assert list(zip(a.tolist(), b.tolist())) == want and len(want) >= N
        cells = ctx.screen_part(refs, 0, 1)                                          # one part: in row order, the form that is added up in LDS
        a2, b2 = ctx.screen_from_cells(refs, cells, 0.8, False)
        assert np.array_equal(a, a2) and np.array_equal(b, b2)
        two = np.concatenate([ctx.screen_part(refs, part, 2) for part in range(2)])  # two parts concatenated: not in row order, the dense matrix
        a3, b3 = ctx.screen_from_cells(refs, two, 0.8, False)
        assert np.array_equal(a, a3) and np.array_equal(b, b3)
    finally:
        refs.close()


def case_marker_set_sizes(ctx):
    """Marker sets of a batch are made by one workgroup per genome in LDS (up to 8192 raw markers per genome), else by device-wide passes: genomes just
    below the capacity, a batch with one genome above it, an empty one."""
    sp = sk.SketchParams(c=30, marker_c=30)                              # every seed is a marker: ~len / 30 raw markers
    for lens in ((100000, 243000, 700), (100000, 262000)):
        genomes = [[("c", random_genome(n, 900 + i))] for i, n in enumerate(lens)]
        names = ["m%d.fa" % i for i in range(len(genomes))]
        ss = ctx.sketch_records(genomes, sp, names)
        for g, rec in enumerate(genomes):
            o = ora.sketch_records(rec, 30, 15, 30, names[g], 1)
            assert np.array_equal(ss.export(g)["markers"], np.sort(o.markers())), (lens, g)
        ss.close()


def fragmented(root, seed, rate, lo=700, hi=9000, drop=0.1):
    """root cut into many contigs of lo..hi bases, a tenth dropped, the rest shuffled, every other one reverse-complemented."""
    rng = np.random.default_rng(seed)
    s = mutate(root, rate, seed + 1)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    ctgs, at = [], 0
    while at < len(s):
        n = int(rng.integers(lo, hi)); piece = s[at:at + n]; at += n
        if rng.random() < drop:
            continue
        if rng.random() < 0.5:
            piece = piece.translate(comp)[::-1]
        ctgs.append(piece)
    order = rng.permutation(len(ctgs))
    return [("ctg%d" % i, ctgs[j]) for i, j in enumerate(order)]


def case_fragmented_genomes(ctx):
    """Draft-assembly shaped inputs: dozens of short contigs in arbitrary order and orientation, so that contig changes
    (chain.rs:786-789), per-contig seed lists (chain.rs:755-780) and both strands occur every few kb on both sides of a pair.
    Exercises the padded-coordinate <-> contig mapping of every stage against the oracle, stage by stage."""
    root = random_genome(260000, 5)
    genomes = [fragmented(root, 10, 0.0), fragmented(root, 20, 0.02), fragmented(root, 30, 0.05, lo=500, hi=3000),
               [("whole", mutate(root, 0.01, 77))], fragmented(random_genome(150000, 6), 40, 0.0)]
    names = ["frag%d.fa" % i for i in range(len(genomes))]
    for mode, c in ((1, 125), (0, 70), (1, 30)):
        ss = ctx.sketch_records(genomes, sk.SketchParams(c=c, seeding_mode=mode), names)
        osk = [ora.sketch_records(g, c, 15, 1000, names[i], mode) for i, g in enumerate(genomes)]
        for g in range(len(genomes)):
            assert_sketch_equal(ss, g, osk[g])
        pr = [0, 0, 0, 1, 1, 2, 3, 0, 2]; pq = [1, 2, 3, 2, 3, 3, 0, 4, 2]
        mp = sk.MapParams(compute_ci=True)
        res, st = ctx.chain_pairs(ss, None, pr, pq, mp, stats=True)
        for x, (i, j) in enumerate(zip(pr, pq)):
            r, s = ora.chain_seeds(osk[i], osk[j], stats=True)
            assert_result_close(res[x], r, (c, i, j))
            assert (int(st[x]["n_anchors"]), int(st[x]["n_qpos"]), int(st[x]["anchor_checksum"]), int(st[x]["n_chunks"]), int(st[x]["n_intervals"]),
                    int(st[x]["n_accepted"]), int(st[x]["n_estimates"])) == \
                (s.n_anchors, s.n_qpos, s.anchor_checksum, s.n_chunks, s.n_intervals, s.n_accepted, s.n_estimates), (c, i, j)
        assert 0.96 < res[0]["ani"] < 0.995 and res[0]["af_ref"] > 0.7 and np.isnan(res[7]["ani"]) and res[8]["ani"] >= 1.0


def case_degenerate_pairs(ctx):
    """empty sketches, unrelated genomes (no anchors -> NaN), both-min-af and min-af cut-offs (-1)."""
    g = [[("a", random_genome(50000, 1))], [("b", random_genome(50000, 2))], [("n", b"N" * 3000)],
         [("a2", mutate(random_genome(50000, 1), 0.01, 3)[:20000])]]
    names = ["d%d" % i for i in range(len(g))]
    ss = ctx.sketch_records(g, sk.SketchParams(), names)
    osk = [ora.sketch_records(x, file_name=names[i]) for i, x in enumerate(g)]
    pr = [0, 0, 2, 0, 3]; pq = [1, 2, 2, 3, 0]
    for kw in (dict(), dict(both_min_af=0.5), dict(min_af=0.9)):
        res = ctx.chain_pairs(ss, None, pr, pq, sk.MapParams(compute_ci=True, **kw))
        for x, (i, j) in enumerate(zip(pr, pq)):
            assert_result_close(res[x], ora.chain_seeds(osk[i], osk[j], **kw), (kw, i, j))
    assert np.isnan(res[0]["ani"]) and np.isnan(res[1]["ani"])


def case_search_resident_db(ctx):
    """search.rs:97-282 against a database resident on the GPU, sharded in two sets; both screening rules; -n; --medium (c=70)."""
    rng = np.random.default_rng(9)
    genomes, names = [], []
    for cl in range(6):
        root = random_genome(int(rng.integers(40000, 70000)), 500 + cl)
        for m in range(5):
            genomes.append([("r%d_%d" % (cl, m), mutate(root, float(rng.uniform(0.002, 0.05)), cl * 17 + m))]); names.append("db%03d.fa" % len(names))
    queries = [[("q%d" % k, mutate(genomes[g][0][1], 0.01, 900 + k))] for k, g in enumerate((0, 7, 14, 29))] + [[("alien", random_genome(50000, 4242))]]
    qnames = ["query%d.fa" % k for k in range(len(queries))]
    for c in (70, 125):
        params = sk.SketchParams(c=c)
        db = sk.build_db(ctx, genomes, params, names, shard_genomes=17)
        assert len(db.shards) == 2 and len(db) == 30
        qs = ctx.sketch_records(queries, params, qnames)
        orefs = [ora.sketch_records(g, c, 15, 1000, names[i]) for i, g in enumerate(genomes)]
        oqs = [ora.sketch_records(g, c, 15, 1000, qnames[i]) for i, g in enumerate(queries)]
        model = ora.Model(MODEL_C125) if c >= 70 else None
        for use_index in (False, True):
            q, r, res = sk.search(ctx, db, qs, use_index=use_index)
            want = {}
            for qi, oq in enumerate(oqs):
                if use_index:
                    cand = [int(x) for x in ora.screen_refs(orefs, oq, 0.8, 2, False)]
                else:
                    cand = [ri for ri in range(len(orefs)) if ora.check_markers_quickly(orefs[ri], oq, 0.8, False)]
                for ri in cand:
                    o = ora.chain_seeds(orefs[ri], oq, min_af=-1.0, model=model)
                    if o.ani > 0.5:
                        want[(qi, ri)] = o
            assert sorted(zip(q.tolist(), r.tolist())) == sorted(want), (c, use_index)
            for a, b, x in zip(q, r, res):
                assert_result_close(x, want[(int(a), int(b))], (c, use_index, int(a), int(b)))
            assert all(res["ani"][i] >= res["ani"][i + 1] for i in range(len(q) - 1) if q[i] == q[i + 1])
            # the oracle's own search loop (ora_search: what bench.py --workload search times as its cpu_baseline) = its pieces put together above
            sq, sr, sres, nch = ora.search(orefs, oqs, 0.8, use_index, min_af=-1.0, model=model, threads=3)
            assert sorted(zip(sq.tolist(), sr.tolist())) == sorted(want) and nch >= len(want)
            for a, b, x in zip(sq, sr, sres):
                o = want[(int(a), int(b))]
                assert x["ani"] == np.float32(o.ani) and x["af_ref"] == np.float32(o.af_ref) and x["af_query"] == np.float32(o.af_query) and x["total_bases_covered"] == o.total_bases_covered
        q1, r1, res1 = sk.search(ctx, db, qs, n_max=2)
        assert max(np.bincount(q1)) <= 2 and not (q1 == 4).any()
        db.close()


def case_large_pair(ctx):
    """A 14 Mbp pair: > 1024 candidate chain intervals => the fallback greedy kernel, several seeding launches' worth of tiles."""
    root = random_genome(14_000_000, 77)
    g = [[("a", root)], [("b1", mutate(root[:9_000_000], 0.12, 5)), ("b2", mutate(root[9_000_000:], 0.12, 6))]]
    names = ["big0.fa", "big1.fa"]
    ss = ctx.sketch_records(g, sk.SketchParams(), names)
    osk = [ora.sketch_records(x, file_name=names[i]) for i, x in enumerate(g)]
    for k in range(2):
        assert_sketch_equal(ss, k, osk[k])
    res, st = ctx.chain_pairs(ss, None, [0, 1], [1, 0], sk.MapParams(compute_ci=True), stats=True)
    for x, (i, j) in enumerate(((0, 1), (1, 0))):
        o, so = ora.chain_seeds(osk[i], osk[j], stats=True)
        assert_result_close(res[x], o, (i, j))
        assert (int(st[x]["n_intervals"]), int(st[x]["n_accepted"]), int(st[x]["n_chunks"]), int(st[x]["anchor_checksum"])) == \
            (so.n_intervals, so.n_accepted, so.n_chunks, so.anchor_checksum)
    assert int(st[0]["n_intervals"]) > 1024


def case_edge_cases_and_errors(ctx):
    """Empty / single-genome sets, genomes without any kept contig, invalid parameters and indices: error codes, never a crash
    (the reference panics or exits; the C ABI must return a status)."""
    import pytest
    # no genomes at all
    ss0 = ctx.sketch_records([], sk.SketchParams(), [])
    assert len(ss0) == 0
    a, b = ctx.screen(ss0, None)
    assert len(a) == 0
    i, j, res, n = ctx.triangle(ss0, sk.MapParams())
    assert len(i) == 0 and n == 0
    # A sketch whose seeds all hash into one narrow stretch of the 32-bit range does not fit the slices of its seed table (a slice of home slots + its
    # slack slots) under the usual hash: the library indexes such a genome again under another salt (common.h table_hash) -- the reference's HashMap
    # takes any key set (types.rs:281-320).  (mix32 is a bijection; these seeds are its preimages of 4000 consecutive hashes.)  The crowded sketch is
    # chained against one that shares every second seed, in both roles, and equals the oracle stage by stage; the same seeds spread over the range likewise.
    def unmix32(h):
        M = 0xFFFFFFFF
        h ^= h >> 16; h = (h * pow(0xc2b2ae35, -1, 1 << 32)) & M
        h ^= (h >> 13) ^ (h >> 26); h = (h * pow(0x85ebca6b, -1, 1 << 32)) & M
        h ^= h >> 16
        return h
    for top, step in ((0xFFFFFFFF, 1), (0x80000000, 1), (0xFFFFFFFF, 1000003)):
        seeds = np.array([unmix32(top - x * step) for x in range(4000)], np.uint32)
        recs = [dict(seed=seeds, pos=np.arange(4000, dtype=np.uint32) * 50, ctgcanon=np.zeros(4000, np.uint32), markers=np.arange(10, dtype=np.uint64),
                     contig_lengths=np.array([250000], np.uint32), total_len=250000),
                dict(seed=seeds[::2], pos=np.arange(2000, dtype=np.uint32) * 100 + 7, ctgcanon=np.ones(2000, np.uint32), markers=np.arange(5, 15, dtype=np.uint64),
                     contig_lengths=np.array([260000], np.uint32), total_len=260000)]
        names = ["crowded.fa", "half.fa"]
        ss = ctx.import_sketches(sk.SketchParams(), recs, names=names)
        assert ss.sizes(0)["n_distinct"] == 4000 and ss.sizes(1)["n_distinct"] == 2000
        osk = [ora.Sketch.from_arrays(125, 15, 1000, names[x], r["seed"], r["pos"], r["ctgcanon"], r["markers"], r["contig_lengths"], r["total_len"]) for x, r in enumerate(recs)]
        res, st = ctx.chain_pairs(ss, None, [0, 1], [1, 0], sk.MapParams(), stats=True)
        for x, (i, j) in enumerate(((0, 1), (1, 0))):
            o, so = ora.chain_seeds(osk[i], osk[j], stats=True)
            assert_result_close(res[x], o, (i, j))
            assert (int(st[x]["n_anchors"]), int(st[x]["n_chunks"]), int(st[x]["n_intervals"]), int(st[x]["anchor_checksum"])) == (so.n_anchors, so.n_chunks, so.n_intervals, so.anchor_checksum)
            assert int(st[x]["n_anchors"]) == 2000
        ss.close()
    # one genome: no pairs; a genome whose contigs are all < 500 bp becomes an empty sketch (chain.rs:618-620 -> NaN)
    g = [[("a", random_genome(20000, 1))], [("short", random_genome(300, 2))]]
    ss = ctx.sketch_records(g, sk.SketchParams(), ["a.fa", "s.fa"])
    assert ss.sizes(1) == dict(n_pos=0, n_distinct=0, n_markers=0, n_contigs=0, total_len=0)
    i, j, res, n = ctx.triangle(ss, sk.MapParams())
    assert len(i) == 0
    r = ctx.chain_pairs(ss, None, [0, 1, 1], [1, 0, 1], sk.MapParams())
    assert np.isnan(r["ani"]).all() and (r["total_bases_covered"] == 0).all()
    one = ctx.sketch_records(g[:1], sk.SketchParams(), ["a.fa"])
    i, j, res, n = ctx.triangle(one, sk.MapParams())
    assert len(i) == 0 and n == 0
    # invalid arguments -> SkaniHipError, context stays usable
    for bad in (sk.SketchParams(k=17), sk.SketchParams(c=2000, marker_c=1000), sk.SketchParams(c=0)):
        with pytest.raises(sk.SkaniHipError):
            ctx.sketch_records(g[:1], bad, ["a.fa"])
    with pytest.raises(sk.SkaniHipError):
        ctx.chain_pairs(ss, None, [5], [0], sk.MapParams())
    with pytest.raises(sk.SkaniHipError):
        ctx.screen(ss, None, 0.8, 7, True)
    with pytest.raises(sk.SkaniHipError):                      # learned ANI needs the model tables
        c2 = sk.Context(0, lib=ctx.L, load_models=False)
        try:
            s2 = c2.sketch_records(g[:1], sk.SketchParams(), ["a.fa"])
            c2.chain_pairs(s2, None, [0], [0], sk.MapParams(learned_ani=True))
        finally:
            c2.close()
    r = ctx.chain_pairs(one, None, [0], [0], sk.MapParams())
    assert r["ani"][0] >= 1.0
    # row-block screen and multi-set chaining: range / set index checks
    with pytest.raises(sk.SkaniHipError):
        ctx.screen_rows(ss, 1, 5)
    with pytest.raises(sk.SkaniHipError):
        ctx.chain_pairs_multi([ss, one], one, [2], [0], [0], sk.MapParams())
    r = ctx.chain_pairs_multi([ss, one], one, [1, 0], [0, 0], [0, 0], sk.MapParams())
    assert r["ani"][0] >= 1.0 and r["ani"][1] >= 1.0
    # a genome whose padded span (length + 8192 per contig) does not fit 31 bits makes its set WIDE (64-bit coordinates): the same sketch comes back
    # out of it, and chained against itself or against the ordinary set it gives the anchors and the ANI of the ordinary pair
    e = one.export(0)
    r0, s0 = ctx.chain_pairs(one, None, [0], [0], sk.MapParams(), stats=True)
    big = dict(e); big["contig_lengths"] = np.array([0x7FFFF000], np.uint32); big["total_len"] = 0x7FFFF000
    many = dict(e); many["contig_lengths"] = np.concatenate([e["contig_lengths"], np.full(300000, 1000, np.uint32)]); many["total_len"] = int(many["contig_lengths"].sum())
    for variant in (big, many):
        wide = ctx.import_sketches(sk.SketchParams(), [variant])
        back = wide.export(0)
        for key in ("seed", "pos", "ctgcanon", "markers", "contig_lengths"):
            assert np.array_equal(back[key], variant[key]), key
        for refs, queries in ((wide, None), (wide, one), (one, wide)):
            r, st = ctx.chain_pairs(refs, queries, [0], [0], sk.MapParams(), stats=True)
            assert r["ani"][0] in (r0["ani"][0], -1.0)                               # (-1: the aligned fraction of a 2 Gbp "genome" is below min_af)
            assert st["n_anchors"][0] == s0["n_anchors"][0] and st["anchor_checksum"][0] == s0["anchor_checksum"][0]
            assert st["n_chunks"][0] == s0["n_chunks"][0] and st["n_accepted"][0] == s0["n_accepted"][0]


def case_database_formats(ctx, tmp):
    """skani database folders through the Python binding: the reference's bundled (pre-0.3) o157 sketch file loads into HBM and
    reproduces the pinned rows; a resident set saved as a database (both flavours) comes back identical, array by array."""
    import os
    from tests.helpers import GOLDEN
    o157, info = sk.load_sketch_files(ctx, [os.path.join(GOLDEN, "e.coli-o157.fasta.sketch")])
    assert info[0]["file_name"] == "test_files/e.coli-o157.fasta" and len(info[0]["contigs"]) == 2 and o157.sizes(0)["n_pos"] == 44127
    names = ["test_files/e.coli-W.fasta", "test_files/o157_plasmid.fasta"]
    refs = ctx.sketch_records([golden_records("e.coli-W.fasta.gz"), golden_records("o157_plasmid.fasta")], sk.SketchParams(), names)
    res = ctx.chain_pairs(refs, o157, [1, 0], [0, 0], sk.MapParams(median=True))
    got = [("%.2f" % (r["ani"] * 100), "%.2f" % (r["af_ref"] * 100), "%.2f" % (r["af_query"] * 100)) for r in res]
    assert got == [("100.00", "99.84", "1.68"), ("98.39", "85.46", "75.97")], got
    w_name = golden_records("e.coli-W.fasta.gz")[0][0]
    infos = [dict(file_name=names[0], contigs=[w_name.decode() if isinstance(w_name, bytes) else w_name], contig_order=0),
             dict(file_name=names[1], contigs=["plasmid"], contig_order=0)]
    for sep in (False, True):
        d = os.path.join(tmp, "db_sep" if sep else "db")
        sk.save_database(refs, d, infos, separate_files=sep)
        back, binfo = sk.load_database(ctx, d)
        assert [i["file_name"] for i in binfo] == names and binfo[1]["contigs"] == ["plasmid"] and (back.params.c, back.params.k, back.params.marker_c) == (125, 15, 1000)
        for g in range(2):
            a, b = refs.export(g), back.export(g)
            for key in ("seed", "pos", "ctgcanon", "markers", "contig_lengths"):
                assert np.array_equal(a[key], b[key]), (sep, g, key)
            assert a["total_len"] == b["total_len"]
        res2 = ctx.chain_pairs(back, o157, [1, 0], [0, 0], sk.MapParams(median=True))
        assert np.array_equal(res2["ani"], res["ani"]) and np.array_equal(res2["af_ref"], res["af_ref"])
    import pytest
    with pytest.raises(FileExistsError):
        sk.save_database(refs, d)
    with pytest.raises(sk.SkaniHipError):
        sk.load_database(ctx, os.path.join(tmp, "nothing_here"))
