"""Shared helpers for the parity tests (fixtures, oracle access)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import oracle_py as ora  # noqa: E402  (tests are allowed to use the oracle)
from skani_amd.fastx import read_fasta  # noqa: E402

MODEL_C125 = os.path.join(ROOT, "skani_amd", "data", "gbdt_c125.bin")
MODEL_C200 = os.path.join(ROOT, "skani_amd", "data", "gbdt_c200.bin")


def pinned():
    return json.load(open(os.path.join(GOLDEN, "pinned.json")))


def golden_records(name):
    return list(read_fasta(os.path.join(GOLDEN, name)))


def o157_arrays():
    z = np.load(os.path.join(GOLDEN, "o157_sketch.npz"))
    return {k: z[k] for k in z.files}


def oracle_o157():
    z = o157_arrays()
    return ora.Sketch.from_arrays(int(z["c"]), int(z["k"]), int(z["marker_c"]), str(z["file_name"]), z["seed"], z["pos"],
                                  z["ctgcanon"], z["markers"], z["contig_lengths"], int(z["total_len"]))


def oracle_sketch_file(name, c=125, k=15, marker_c=1000, mode=1, file_name=None):
    return ora.sketch_records(golden_records(name), c, k, marker_c, file_name or ("test_files/" + name), mode)


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def random_genome(length, seed, n_rate=0.0):
    rng = np.random.default_rng(seed)
    s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, length)]
    if n_rate > 0:
        s = s.copy(); s[rng.random(length) < n_rate] = ord("N")
    return s.tobytes()


def big_random_genome(length, seed, piece=1 << 26):
    """random_genome for lengths in the Gbp range: made piece by piece from one-byte draws (random_genome's int64 draws take 8 bytes per base)"""
    rng = np.random.default_rng(seed); acgt = np.frombuffer(b"ACGT", np.uint8)
    out = np.empty(length, np.uint8)
    for at in range(0, length, piece):
        n = min(piece, length - at)
        out[at:at + n] = acgt[rng.integers(0, 4, n, dtype=np.uint8)]
    return out


def big_mutate(a, rate, seed, piece=1 << 26):
    """substitutions at `rate` in a uint8 array of bases, piece by piece; returns a new array"""
    rng = np.random.default_rng(seed); acgt = np.frombuffer(b"ACGT", np.uint8)
    lut = np.zeros(256, np.uint8); lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
    out = a.copy()
    for at in range(0, len(a), piece):
        v = out[at:at + piece]
        idx = np.nonzero(rng.random(len(v), dtype=np.float32) < rate)[0]
        v[idx] = acgt[(lut[v[idx]] + rng.integers(1, 4, len(idx), dtype=np.uint8)) % 4]
    return out


def mutate(seq, rate, seed):
    rng = np.random.default_rng(seed)
    a = np.frombuffer(seq, np.uint8).copy()
    idx = np.nonzero(rng.random(len(a)) < rate)[0]
    lut = np.zeros(256, np.uint8); lut[ord("A")] = 0; lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
    codes = lut[a[idx]]
    codes = (codes + rng.integers(1, 4, len(idx))) % 4
    a[idx] = np.frombuffer(b"ACGT", np.uint8)[codes]
    return a.tobytes()
