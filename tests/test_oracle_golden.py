"""Pins the CPU oracle against every reference golden vector whose inputs exist in the checkout
(SURVEY.md section 8c).  CPU-only."""
import numpy as np
import pytest

from tests.helpers import (MODEL_C125, golden_records, o157_arrays, ora, oracle_o157, oracle_sketch_file, pinned)


def test_mm_hash_known_answers():
    """types.rs:86-96; every key of the golden sketch passes h < u64::MAX/125."""
    z = o157_arrays()
    thr = (2**64 - 1) // 125
    keys = np.unique(z["seed"])
    assert len(keys) == 40716
    hs = [ora.lib().ora_mm_hash64(int(k)) for k in keys[::37]]
    assert all(h < thr for h in hs)
    assert ora.lib().ora_mm_hash64(0) == 0x77CFA1EEF01BCA90  # mix of 0, independent hand computation


@pytest.mark.parametrize("mode", [0, 1])
def test_plasmid_seeds_match_golden_sketch(mode):
    """Contig 1 of the reference's e.coli-o157.fasta.sketch is exactly o157_plasmid.fasta: 759 seed records and
    81 markers must be reproduced bit-exactly by both seeding modes ((92596-20)%4==0, no N)."""
    z = o157_arrays()
    sel = (z["ctgcanon"] >> 1) == 1
    gold = np.stack([z["seed"][sel], z["pos"][sel], z["ctgcanon"][sel] & 1], 1)
    assert len(gold) == 759
    sk = oracle_sketch_file("o157_plasmid.fasta", mode=mode)
    s, p, cc = sk.seeds()
    mine = np.stack([s, p, cc & 1], 1)
    assert np.array_equal(mine, gold)
    assert sk.n_markers == 81
    assert np.isin(sk.markers(), z["markers"]).all()
    assert sk.total_len == 92596 and list(sk.contig_lengths()) == [92596]


def _pct(x):
    return round(float(x) * 100 + 1e-9, 2)


def test_pinned_triples_median():
    """test_results_versions/0.3.0:130-135 (search --median: learned ANI off)."""
    o157 = oracle_o157()
    refs = {"o157_plasmid.fasta": oracle_sketch_file("o157_plasmid.fasta"),
            "e.coli-W.fasta.gz": oracle_sketch_file("e.coli-W.fasta.gz", file_name="test_files/e.coli-W.fasta"),
            "o157_sketch": o157}
    for t in pinned()["triples_median_percent"]:
        r = ora.chain_seeds(refs[t["ref"]], o157, median=True)
        got = ("%.2f" % (r.ani * 100), "%.2f" % (r.af_ref * 100), "%.2f" % (r.af_query * 100))
        want = ("%.2f" % t["ani"], "%.2f" % t["af_ref"], "%.2f" % t["af_query"])
        assert got == want, (t, got)


def test_w_vs_w():
    """tests/tests.rs:42-60."""
    w = oracle_sketch_file("e.coli-W.fasta.gz")
    assert (w.n_positions, w.n_distinct, w.n_markers) == (39310, 37786, 4649)   # SURVEY section 8 measurements
    r = ora.chain_seeds(w, w)
    assert r.ani >= 1.0 and r.af_query >= 0.99 and r.af_ref >= 0.99


def test_w_vs_o157_stage_counts():
    """Stage sizes measured at survey time (SURVEY section 8 header)."""
    w = oracle_sketch_file("e.coli-W.fasta.gz", file_name="test_files/e.coli-W.fasta")
    r, st = ora.chain_seeds(w, oracle_o157(), median=True, stats=True)
    assert (st.n_anchors, st.n_chunks, st.n_intervals, st.n_accepted, st.n_estimates) == (31989, 245, 500, 380, 234)


def test_viruses_individual_contigs():
    """tests/int_test_new.rs:57-62: triangle -i over viruses.fna -> one ANI in (99.0,99.9), another > 99.9.
    All three contigs have (L-20)%4 != 0 => exercises the avx2 tail rule."""
    recs = golden_records("viruses.fna")
    sks = []
    for j, (name, seq) in enumerate(recs):
        sk = ora.Sketch(125, 15, 1000, "test_files/viruses.fna"); sk.add_contig(seq, 1); sks.append(sk)
    anis = []
    for i in range(len(sks)):
        for j in range(i + 1, len(sks)):
            r = ora.chain_seeds(sks[i], sks[j])
            if r.ani > 0.1:
                anis.append(r.ani * 100)
    assert any(99.0 < a < 99.9 for a in anis) and any(a > 99.9 for a in anis), anis


def test_avx2_equals_scalar_on_120bp_literal():
    """tests/tests.rs:130-144."""
    p = pinned()["avx2_vs_scalar_120bp"]
    a = ora.Sketch(p["c"], 15, 1000, ""); a.add_contig(p["seq"].encode(), 1, 0)
    b = ora.Sketch(p["c"], 15, 1000, ""); b.add_contig(p["seq"].encode(), 0, 0)
    assert a.n_positions > 0
    for x, y in zip(a.seeds(), b.seeds()):
        assert np.array_equal(x, y)
    assert np.array_equal(a.markers(), b.markers())


def test_all_n():
    """tests/tests.rs:149-157 and int_test_new.rs:157-161."""
    p = pinned()["all_n_150bp"]
    sk = ora.Sketch(p["c"], 15, 1000, ""); sk.add_contig(p["seq"].encode(), 0, 0)
    assert sk.n_distinct == 0
    recs = golden_records("all_ns.fa")
    sk = ora.sketch_records(recs, mode=1)
    assert sk.n_positions == 0


def test_learned_ani_lowers_or_keeps(tmp_path):
    """tests/tests.rs:121-126 asserts learned <= raw on missing inputs; here on W vs o157 (unpinned value)."""
    m = ora.Model(MODEL_C125)
    w = oracle_sketch_file("e.coli-W.fasta.gz", file_name="test_files/e.coli-W.fasta"); o = oracle_o157()
    raw = ora.chain_seeds(w, o); learned = ora.chain_seeds(w, o, model=m)
    assert 0.97 < learned.ani <= 1.0 and abs(learned.ani - raw.ani) < 0.01
    assert learned.af_ref == raw.af_ref


def test_screen_rules():
    """screen.rs:84-189 on the W/o157/plasmid trio (W and o157 share 2,972 markers; cutoff 42)."""
    w = oracle_sketch_file("e.coli-W.fasta.gz"); o = oracle_o157(); p = oracle_sketch_file("o157_plasmid.fasta")
    assert len(np.intersect1d(w.markers(), o.markers())) == 2972
    assert ora.check_markers_quickly(w, o, 0.8, False)
    assert list(ora.screen_refs([w, o, p], o, 0.8, 0, True)) == [0, 1, 2]
    assert abs(ora.lib().ora_powi(0.8, 21) - 0.8**21) < 1e-15


def test_avx2_intrinsics_equal_their_plain_statement():
    """The oracle's mode 1 runs avx2_seeding.rs's algorithm with AVX2 intrinsics (what bench.py's CPU baseline times); mode 2 is the plain C++
    statement of the same lines.  Same seeds, positions, strands and markers on the golden plasmid, E. coli W, the viruses (tail rule) and on random
    contigs with N runs, lower-case bases and every length class mod 4; sketch_batch (threads) equals per-contig calls."""
    from tests.helpers import random_genome
    rng = np.random.default_rng(77)
    sets = [golden_records("o157_plasmid.fasta"), golden_records("e.coli-W.fasta.gz"), golden_records("viruses.fna")]
    rnd = []
    for i in range(12):
        s = bytearray(random_genome(int(rng.integers(600, 60000)) + i % 4, 500 + i, n_rate=0.0005 if i % 3 == 0 else 0.0))
        if i % 2:
            a = int(rng.integers(0, len(s) - 300)); s[a:a + 200] = bytes(s[a:a + 200]).lower()
        if i % 5 == 0:
            a = int(rng.integers(0, len(s) - 300)); s[a:a + 120] = b"N" * 120
        rnd.append(("r%d" % i, bytes(s)))
    sets.append(rnd)
    for c in (125, 30):
        fast = ora.sketch_batch(sets, c, 15, 1000, ["f%d" % i for i in range(len(sets))], 1, 500, 3)
        for g, recs in enumerate(sets):
            plain = ora.sketch_records(recs, c, 15, 1000, "f%d" % g, 2)
            one = ora.sketch_records(recs, c, 15, 1000, "f%d" % g, 1)
            for sk in (fast[g], one):
                for u, v in zip(sk.seeds(True), plain.seeds(True)):
                    assert np.array_equal(u, v)
                assert np.array_equal(sk.markers(), plain.markers()) and list(sk.contig_lengths()) == list(plain.contig_lengths())
            assert plain.n_positions > 0
