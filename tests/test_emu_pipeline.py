"""Runs the parity cases against the kernel sources compiled for the CPU simulator (tests/emu).  This is a
debugging aid for a container without a GPU -- NOT a parity claim (those are the -m gpu tests)."""
import numpy as np
import pytest

import skani_amd as sk
from tests import parity_cases as pc
from tests.emu_lib import emu_lib


@pytest.fixture(scope="module", params=["narrow", "wide"])
def ctx(request):
    """Every case runs twice: with ordinary sketch sets, and with SKH_TUNE_WIDE_SPAN=0, which makes every sketch set a wide one (64-bit coordinates:
    the path of genomes beyond 2^31 padded bases, include/skani_hip.h skh_sketch_is_wide).  Tunables are read when the context is made."""
    import os
    if request.param == "wide": os.environ["SKH_TUNE_WIDE_SPAN"] = "0"
    try:
        c = sk.Context(0, lib=emu_lib())
    finally:
        os.environ.pop("SKH_TUNE_WIDE_SPAN", None)
    probe = c.sketch_records([[("a", pc.random_genome(3000, 1))]], sk.SketchParams(), ["a.fa"])
    assert probe.wide == (request.param == "wide")
    probe.close()
    yield c
    c.close()


def test_emu_seeding_golden(ctx): pc.case_seeding_golden_plasmid(ctx)
def test_emu_seeding_fixtures(ctx): pc.case_seeding_fixtures(ctx)
def test_emu_pack_every_byte(ctx): pc.case_pack_every_byte(ctx)
def test_emu_pack_in_batches(ctx): pc.case_pack_in_batches(ctx)
def test_emu_seeding_ecoli_w(ctx): pc.case_seeding_ecoli_w(ctx)
def test_emu_seeding_low_complexity(ctx): pc.case_seeding_low_complexity(ctx)
def test_emu_pinned_triples(ctx): pc.case_pinned_triples(ctx)
def test_emu_w_vs_w(ctx): pc.case_w_vs_w(ctx)
def test_emu_viruses(ctx): pc.case_viruses_individual(ctx)
def test_emu_triangle_synthetic(ctx): pc.case_triangle_synthetic(ctx, params=((1, 125), (0, 30), (1, 20)), length=120000)
def test_emu_triangle_dense_sketches(ctx): pc.case_triangle_synthetic(ctx, params=((1, 8), (0, 3)), length=30000)   # c < 10: chain band beyond 256 anchors
def test_emu_screen_rules(ctx): pc.case_screen_rules(ctx)
def test_emu_screen_marker_prefix_groups(ctx): pc.case_screen_marker_prefix_groups(ctx)
def test_emu_marker_set_sizes(ctx): pc.case_marker_set_sizes(ctx)
def test_emu_degenerate(ctx): pc.case_degenerate_pairs(ctx)
def test_emu_fragmented_genomes(ctx): pc.case_fragmented_genomes(ctx)
def test_emu_database_formats(ctx, tmp_path): pc.case_database_formats(ctx, str(tmp_path))
def test_emu_search_resident_db(ctx): pc.case_search_resident_db(ctx)
def test_emu_large_pair(ctx): pc.case_large_pair(ctx)
def test_emu_edge_cases(ctx): pc.case_edge_cases_and_errors(ctx)


def test_emu_small_budgets_force_multi_batch_paths(monkeypatch):
    """Shrink the scratch budgets so that tiny inputs go through several seeding launches (a genome's tiles split across
    launches), several screen row blocks, several chain super-batches and batches."""
    monkeypatch.setenv("SKH_TUNE_SEED_SCRATCH_BYTES", "6000")        # ~2 tiles per launch
    monkeypatch.setenv("SKH_TUNE_SCREEN_CELLS", "20")                 # 2 rows per block at 7 genomes
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
    c = sk.Context(0, lib=emu_lib())
    try:
        pc.case_triangle_synthetic(c, params=((1, 125),), length=60000)
    finally:
        c.close()
    monkeypatch.delenv("SKH_TUNE_SCAN_TWO_MAX")
    c = sk.Context(0, lib=emu_lib())
    try:
        pc.case_triangle_synthetic(c, params=((1, 125), (0, 30)), length=60000)
        pc.case_screen_rules(c)
        pc.case_seeding_fixtures(c)
    finally:
        c.close()


def test_emu_selection_in_global_memory(monkeypatch):
    """SKH_TUNE_GREEDY_BIG_MIN=2: every pair with two or more candidate intervals selects its chains with greedy_big_kernel (the kernel of pairs with
    thousands of candidates: its sort, its bin lists and its long-interval list in global memory)."""
    import importlib.util, os
    monkeypatch.setenv("SKH_TUNE_GREEDY_BIG_MIN", "2")
    c = sk.Context(0, lib=emu_lib())
    try:
        pc.case_large_pair(c)
        pc.case_fragmented_genomes(c)
        pc.case_triangle_synthetic(c, params=((1, 125), (0, 30)), length=150000)
        spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
        fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
        rng = np.random.default_rng(99)
        assert sum(fz.one_round(c, rng, r) for r in range(6)) > 30
    finally:
        c.close()


def test_emu_sets_of_both_kinds_of_genomes(monkeypatch):
    """SKH_TUNE_WIDE_SPAN between the sizes of the fuzzer's genomes: sets with wide and ordinary genomes side by side, chaining calls that split into an
    ordinary and a 64-bit run."""
    import importlib.util, os
    monkeypatch.setenv("SKH_TUNE_WIDE_SPAN", "120000")
    c = sk.Context(0, lib=emu_lib())
    try:
        spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
        fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
        rng = np.random.default_rng(123)
        assert sum(fz.one_round(c, rng, r) for r in range(8)) > 40
    finally:
        c.close()


def test_emu_randomised_differential(ctx):
    """a few rounds of tools/fuzz_parity.py through the simulator"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    rng = np.random.default_rng(7)
    assert sum(fz.one_round(ctx, rng, r) for r in range(4)) > 20


def test_emu_seeding_drops_candidates_that_are_not_hits():
    """The seeding loop lists a SUPERSET of the hits (32-bit compare on the hash's leading word, pack_seed.hip seed_probe) and its dense pass drops the
    rest.  The simulator's superset is deliberately loose, so the drop path -- one candidate in ~2^31 on the GPU -- runs here on ordinary sequence,
    in both forms (a tile that lists its hits, and a low-complexity tile that only counts because it overflows the capped scratch), and the sketches
    still equal the oracle's."""
    import ctypes
    from tests.helpers import ora
    import os
    L = emu_lib()
    drops = ctypes.c_ulonglong.in_dll(L, "skh_emu_seed_drops")
    g = pc.random_genome(600000, 99)
    low = np.frombuffer((b"ACGTTGCA" * 3000), np.uint8).copy()                # few distinct seeds: a tile beyond the capped scratch
    for tile_cap in (None, "8"):                                               # SKH_TUNE_SEED_TILE_CAP=8: every tile overflows the capped scratch, counts only, and is listed by the re-run
        if tile_cap: os.environ["SKH_TUNE_SEED_TILE_CAP"] = tile_cap
        try:
            c = sk.Context(0, lib=L)
        finally:
            os.environ.pop("SKH_TUNE_SEED_TILE_CAP", None)
        try:
            before = drops.value
            for mode in (sk.SEED_SCALAR, sk.SEED_AVX2):
                for cc in (125, 2):
                    recs = [[("a", g[:200000 if cc == 2 else len(g)])], [("l", low)]]
                    ss = c.sketch_records(recs, sk.SketchParams(cc, 15, 1000, mode), ["a.fa", "l.fa"])
                    for k, r in enumerate(recs):
                        o = ora.sketch_records(r, c=cc, k=15, marker_c=1000, mode=mode, file_name="x.fa")
                        e = ss.export(k); s, p, q = o.seeds(pos_order=True)
                        assert np.array_equal(e["seed"], s) and np.array_equal(e["pos"], p) and np.array_equal(e["ctgcanon"], q) and np.array_equal(e["markers"], o.markers())
                    ss.close()
            assert drops.value - before > (40 if tile_cap else 20), (tile_cap, drops.value - before)   # (with the small cap every drop happens twice: counting, then listing)
        finally:
            c.close()


def test_emu_reference_copies_2_to_the_32_apart(monkeypatch):
    """The ring DP of a 64-bit run compares LOW words and accepts a link only when the 64-bit difference of the reference coordinates has a zero high word
    (chain_dp.h).  A simulator build with contigs 2^30 - 5450 padded coordinates apart (tests/emu/build_emu.py "bigpad") makes the case that needs the high words
    out of a few kilobases: the reference holds a 20 kb contig A, three short contigs, and A again -- the second copy starts EXACTLY 2^32 coordinates behind the
    first (4 paddings + 20,000 + 3 x 600 bases), so every anchor on the copy has the low word of its twin on the original: a DP that looked at low words only would
    chain across the two copies (gap 0, a tie the larger index wins).  Against the oracle, whose coordinates are (contig, position); through the sweep kernel too."""
    from tests.helpers import MODEL_C125, mutate, ora, random_genome
    from tests.parity_cases import assert_result_close
    A = random_genome(20000, 911)
    ref = [("A", A), ("f1", random_genome(600, 912)), ("f2", random_genome(600, 913)), ("f3", random_genome(600, 914)), ("A2", A), ("tail", random_genome(30000, 915))]
    # (eight short contigs keep the query's mean contig length, hence its switch_qr score, below the reference's: the reference is the PROBED side, chain.rs:15-26)
    qry = [("qa", mutate(A, 0.01, 916)), ("qt", mutate(ref[5][1], 0.02, 917))] + [("s%d" % i, random_genome(600, 930 + i)) for i in range(8)]
    names = ["ref.fa", "qry.fa"]
    osk = [ora.sketch_records(ref, file_name=names[0]), ora.sketch_records(qry, file_name=names[1])]
    want, wst = ora.chain_seeds(osk[0], osk[1], model=ora.Model(MODEL_C125), stats=True)
    assert wst.n_anchors > 300 and wst.n_accepted >= 2 and not wst.switched
    got = []
    for sweep in ("0", "1"):
        monkeypatch.setenv("SKH_TUNE_WIDE_SWEEP_DP", sweep)
        c = sk.Context(0, lib=emu_lib("bigpad"))
        try:
            ss = c.sketch_records([ref, qry], sk.SketchParams(), names)
            assert ss.wide
            for g in range(2):
                pc.assert_sketch_equal(ss, g, osk[g])
            res, st = c.chain_pairs(ss, None, [0], [1], sk.MapParams(learned_ani=True, compute_ci=True), stats=True)
            assert_result_close(res[0], want, ("bigpad", sweep))
            assert (int(st[0]["n_anchors"]), int(st[0]["n_chunks"]), int(st[0]["n_intervals"]), int(st[0]["n_accepted"]), int(st[0]["n_estimates"]), int(st[0]["anchor_checksum"])) == \
                (wst.n_anchors, wst.n_chunks, wst.n_intervals, wst.n_accepted, wst.n_estimates, wst.anchor_checksum), sweep
            got.append(res[0].tobytes())
        finally:
            c.close()
    assert got[0] == got[1]


def test_emu_screen_incidence_sort(monkeypatch):
    def make_ctx(env):
        import os
        for k, v in env.items(): os.environ[k] = v
        try: return sk.Context(0, lib=emu_lib())
        finally:
            for k in env: os.environ.pop(k, None)
    pc.case_screen_incidence_sort(make_ctx)


def test_emu_screen_count_walks(monkeypatch):
    def make_ctx(env):
        import os
        for k, v in env.items(): os.environ[k] = v
        try: return sk.Context(0, lib=emu_lib())
        finally:
            for k in env: os.environ.pop(k, None)
    pc.case_screen_count_walks(make_ctx, G=800)              # (800 genomes share the universal marker: its group still runs past a tile and its halo on both sides)


def test_emu_every_genome_resalted(monkeypatch):
    """SKH_TUNE_BUILD_RESALT_ALL=1: the table build indexes every genome a second time under salt 1 -- the path of a genome whose seeds crowd a stretch of the hash
    range, which no ordinary genome takes -- after the sketch call has made the per-genome tables of the pair descriptors AHEAD of the build's end (with salt 0):
    they must be made again, or the join probes with the wrong hash and finds nothing."""
    monkeypatch.setenv("SKH_TUNE_BUILD_RESALT_ALL", "1")
    c = sk.Context(0, lib=emu_lib())
    try:
        pc.case_triangle_synthetic(c, params=((1, 125),), length=60000)
    finally:
        c.close()


@pytest.mark.parametrize("env", [{"SKH_TUNE_SKEYS_AVG": "16"}, {"SKH_TUNE_SKEYS_AVG": "16", "SKH_TUNE_SKEYS_CAP": "16"}, {"SKH_TUNE_SCREEN_SORT_RADIX": "1"},
                                 {"SKH_TUNE_SCREEN_COL_ORDER": "0"}, {"SKH_TUNE_SCREEN_COL_ORDER": "2"}, {"SKH_TUNE_SCREEN_COUNT_ROWS": "0"}, {"SKH_TUNE_MARKER_GATE": "0"}])   # (the last three: the count kernel without its column order / with it also in small key-range parts / the count kernel of rounds 1-5; then the sketch call without the marker sets' head start)
def test_emu_incidence_sort_through_the_sketch_call(monkeypatch, env):
    """The sketch call's own way through the incidence sort -- buckets counted beside the marker sets, their largest read back with the set sizes, keys placed after the
    gather, the last kernel not waited for -- with tiny buckets, with a capacity the largest bucket exceeds (the radix sort takes the bucketed keys, in the set's own
    scratch) and with the radix sort alone; the triangle's candidates and results against the oracle."""
    for k, v in env.items(): monkeypatch.setenv(k, v)
    c = sk.Context(0, lib=emu_lib())
    try:
        pc.case_triangle_synthetic(c, params=((1, 125),), length=60000)
        pc.case_screen_rules(c)
    finally:
        c.close()
