#!/usr/bin/env python3
"""Randomised differential test: GPU path (through the C ABI) vs the CPU oracle on many small random genome pairs.
usage (on a GPU box): python tools/fuzz_parity.py [n_rounds] [seed] [big]     -- prints a summary line; exits 1 on the first mismatch.
Covers what the fixed parity cases only touch in a few places: repeats (tandem / dispersed duplications), inversions, N runs,
many short contigs, contigs below the 500-bp cut, all compression factors / k / both seeding semantics, robust / median / CI."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skani_amd as sk                                        # noqa: E402
from tests import parity_cases as pc                          # noqa: E402
from tests.helpers import MODEL_C125, MODEL_C200, mutate, ora, random_genome   # noqa: E402

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def scramble(seq, rng):
    """structural edits: tandem / dispersed duplications, inversions, deletions, N runs"""
    s = bytearray(seq)
    for _ in range(int(rng.integers(0, 6))):
        op = int(rng.integers(0, 5)); n = len(s)
        if n < 2000:
            break
        a = int(rng.integers(0, n - 1000)); ln = int(rng.integers(50, min(20000, n - a)))
        piece = bytes(s[a:a + ln])
        if op == 0:
            s[a:a] = piece * int(rng.integers(1, 4))                          # tandem duplication
        elif op == 1:
            b = int(rng.integers(0, n)); s[b:b] = piece                        # dispersed copy
        elif op == 2:
            s[a:a + ln] = piece.translate(COMP)[::-1]                          # inversion
        elif op == 3:
            del s[a:a + ln]                                                    # deletion
        else:
            s[a:a + min(ln, 300)] = b"N" * min(ln, 300)                        # N run
    return bytes(s)


def contigs_of(seq, rng):
    k = int(rng.choice([1, 1, 2, 5, 30]))
    if k == 1 or len(seq) < 4000:
        return [("c0", seq)]
    cuts = np.sort(rng.integers(1, len(seq) - 1, k - 1))
    out, prev = [], 0
    for i, c in enumerate(list(cuts) + [len(seq)]):
        piece = seq[prev:c]; prev = c
        if rng.random() < 0.3:
            piece = piece.translate(COMP)[::-1]
        out.append(("c%d" % i, piece))
    order = rng.permutation(len(out))
    return [out[i] for i in order]


def one_round(ctx, rng, rnd, big=False):
    c = int(rng.choice([20, 30, 70, 125, 125, 200])); k = int(rng.choice([14, 15, 15, 16])); m = int(rng.choice([200, 1000])); mode = int(rng.integers(0, 2))
    if c > m:
        m = 1000
    n_genomes = int(rng.integers(1, 4)) if big else int(rng.integers(2, 7))
    root = random_genome(int(rng.integers(400000, 4000000)) if big else int(rng.integers(3000, 250000)), int(rng.integers(0, 2**31)))
    genomes = []
    for g in range(n_genomes):
        base = root if rng.random() < 0.8 else random_genome(int(rng.integers(1000, 100000)), int(rng.integers(0, 2**31)))
        s = mutate(base, float(rng.choice([0.0, 0.002, 0.01, 0.03, 0.08, 0.15])), int(rng.integers(0, 2**31)))
        genomes.append(contigs_of(scramble(s, rng), rng))
    names = ["f%02d_%d.fa" % (rnd % 7, g) for g in range(n_genomes)]
    ss = ctx.sketch_records(genomes, sk.SketchParams(c, k, m, mode), names)
    osk = [ora.sketch_records(g, c, k, m, names[i], mode) for i, g in enumerate(genomes)]
    for g in range(n_genomes):
        pc.assert_sketch_equal(ss, g, osk[g])
    pr = [i for i in range(n_genomes) for j in range(n_genomes)]; pq = [j for i in range(n_genomes) for j in range(n_genomes)]
    kw = [dict(), dict(robust=True), dict(median=True)][int(rng.integers(0, 3))]
    learned = sk.use_learned_ani(c) and not kw.get("median")
    model = ora.Model(MODEL_C125 if abs(c - 125) < abs(c - 200) else MODEL_C200) if learned else None
    res, st = ctx.chain_pairs(ss, None, pr, pq, sk.MapParams(learned_ani=learned, compute_ci=True, **kw), stats=True)
    for x, (i, j) in enumerate(zip(pr, pq)):
        o, so = ora.chain_seeds(osk[i], osk[j], model=model, stats=True, **kw)
        pc.assert_result_close(res[x], o, (rnd, c, k, m, mode, kw, i, j))
        got = (int(st[x]["n_anchors"]), int(st[x]["n_qpos"]), int(st[x]["anchor_checksum"]), int(st[x]["n_chunks"]), int(st[x]["n_intervals"]),
               int(st[x]["n_accepted"]), int(st[x]["n_estimates"]))
        want = (so.n_anchors, so.n_qpos, so.anchor_checksum, so.n_chunks, so.n_intervals, so.n_accepted, so.n_estimates)
        assert got == want, (rnd, c, k, m, mode, kw, i, j, got, want)
    # the screens: triangle form, and the three query-vs-reference rules with this set on both sides, at a random cut-off
    ident = float(rng.choice([0.0, 0.8, 0.9, 0.97])); rescue = bool(rng.integers(0, 2)); eff = ident if ident else 0.8
    a, b = ctx.screen(ss, None, ident, 0, rescue)
    exp = sorted((i, int(j)) for i in range(n_genomes - 1) for j in ora.screen_refs(osk, osk[i], eff, 0, rescue) if j > i)
    assert list(zip(a.tolist(), b.tolist())) == exp, (rnd, "screen tri", ident, rescue)
    for rule in (0, 2):
        a, b = ctx.screen(ss, ss, ident, rule, rescue)
        exp = sorted((q, int(r)) for q in range(n_genomes) for r in ora.screen_refs(osk, osk[q], eff, rule, rescue))
        assert list(zip(a.tolist(), b.tolist())) == exp, (rnd, "screen rule", rule, ident, rescue)
    a, b = ctx.screen(ss, ss, ident, 1, rescue)
    exp = sorted((q, r) for q in range(n_genomes) for r in range(n_genomes) if ora.check_markers_quickly(osk[r], osk[q], eff, rescue))
    assert list(zip(a.tolist(), b.tolist())) == exp, (rnd, "screen quick", ident, rescue)
    return len(pr)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == "big"       # Mbp-sized genomes, 1-3 per set (mid-sized sorts, many tiles per pair)
    lib = None
    if os.environ.get("SKANI_FUZZ_EMU"):
        from tests.emu_lib import emu_lib
        lib = emu_lib()
    ctx = sk.Context(0, lib=lib) if lib else sk.Context(0)
    rng = np.random.default_rng(seed)
    pairs = 0
    for r in range(rounds):
        try:
            pairs += one_round(ctx, rng, r, big)
        except AssertionError as e:
            print("MISMATCH in round %d (seed %d): %s" % (r, seed, str(e)[:600]))
            sys.exit(1)
    print("fuzz ok: %d rounds, %d pairs, seed %d" % (rounds, pairs, seed))


if __name__ == "__main__":
    main()
