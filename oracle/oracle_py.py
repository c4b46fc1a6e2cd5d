"""ctypes binding of the CPU oracle (oracle/libskani_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by the skani_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class AniResult(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("ani", "af_query", "af_ref", "ci_lower", "ci_upper", "std",
                 "q90_q", "q90_r", "q50_q", "q50_r", "q10_q", "q10_r")] + \
               [(n, C.c_uint32) for n in ("num_contigs_q", "num_contigs_r", "avg_chain_int_len", "total_bases_covered")]


class MapOpts(C.Structure):
    _fields_ = [("min_af", C.c_double), ("both_min_af", C.c_double), ("robust", C.c_int), ("median", C.c_int)]


class ChainStats(C.Structure):
    _fields_ = [("switched", C.c_int)] + [(n, C.c_uint64) for n in
                ("n_anchors", "n_chunks", "n_intervals", "n_accepted", "n_estimates", "n_qpos",
                 "anchor_checksum", "interval_checksum")]


RESULT_DTYPE = np.dtype([(n, np.float32) for n in
                         ("ani", "af_query", "af_ref", "ci_lower", "ci_upper", "std",
                          "q90_q", "q90_r", "q50_q", "q50_r", "q10_q", "q10_r")] +
                        [(n, np.uint32) for n in ("num_contigs_q", "num_contigs_r", "avg_chain_int_len",
                                                  "total_bases_covered")])


def build(force=False):
    so = os.path.join(_HERE, "libskani_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("skani_oracle.cpp", "skani_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
        L.ora_sketch_new.restype = vp; L.ora_sketch_new.argtypes = [u32, u32, u32, C.c_char_p]
        L.ora_sketch_free.argtypes = [vp]
        L.ora_sketch_add_contig.restype = i32; L.ora_sketch_add_contig.argtypes = [vp, vp, u64, i32, u64]
        L.ora_sketch_batch.restype = None
        L.ora_sketch_batch.argtypes = [u32, vp, vp, vp, u32, u32, u32, vp, i32, u64, i32, vp]
        L.ora_sketch_files.restype = None; L.ora_sketch_files.argtypes = [u32, vp, u32, u32, u32, i32, u64, i32, vp]
        L.ora_sketch_from_arrays.restype = vp
        L.ora_sketch_from_arrays.argtypes = [u32, u32, u32, C.c_char_p, vp, vp, vp, u64, vp, u64, vp, u32, u64]
        for n in ("n_positions", "n_distinct", "n_markers", "total_len"):
            f = getattr(L, "ora_sketch_" + n); f.restype = u64; f.argtypes = [vp]
        L.ora_sketch_n_contigs.restype = u32; L.ora_sketch_n_contigs.argtypes = [vp]
        L.ora_sketch_export_seeds.argtypes = [vp, vp, vp, vp]
        L.ora_sketch_export_seeds_pos_order.argtypes = [vp, vp, vp, vp]
        L.ora_sketch_export_markers.argtypes = [vp, vp]
        L.ora_sketch_export_contig_lengths.argtypes = [vp, vp]
        L.ora_model_load.restype = vp; L.ora_model_load.argtypes = [C.c_char_p]
        L.ora_model_free.argtypes = [vp]
        L.ora_model_predict.restype = C.c_float; L.ora_model_predict.argtypes = [vp, vp]
        L.ora_chain_seeds.argtypes = [vp, vp, C.POINTER(MapOpts), vp, C.POINTER(AniResult), C.POINTER(ChainStats)]
        L.ora_check_markers_quickly.restype = i32; L.ora_check_markers_quickly.argtypes = [vp, vp, dbl, i32]
        L.ora_screen_refs.restype = u64; L.ora_screen_refs.argtypes = [vp, u32, vp, dbl, i32, i32, vp]
        L.ora_triangle.restype = u64
        L.ora_triangle.argtypes = [vp, u32, dbl, i32, C.POINTER(MapOpts), vp, i32, vp, vp, vp, u64, vp, vp]
        L.ora_search.restype = u64
        L.ora_search.argtypes = [vp, u32, vp, u32, dbl, i32, C.POINTER(MapOpts), vp, i32, vp, vp, vp, u64, vp]
        L.ora_triangle_phases.restype = None; L.ora_triangle_phases.argtypes = [C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]
        L.ora_mm_hash64.restype = u64; L.ora_mm_hash64.argtypes = [u64]
        L.ora_powi.restype = dbl; L.ora_powi.argtypes = [dbl, i32]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Sketch:
    """One genome sketch held by the oracle (types.rs:252-277)."""

    def __init__(self, c=125, k=15, marker_c=1000, file_name="", handle=None):
        self.c, self.k, self.marker_c, self.file_name = c, k, marker_c, file_name
        self.h = handle if handle is not None else lib().ora_sketch_new(c, k, marker_c, file_name.encode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_sketch_free(self.h); self.h = None

    def add_contig(self, seq: bytes, mode=1, min_len=500):
        buf = np.frombuffer(seq, dtype=np.uint8)
        return lib().ora_sketch_add_contig(self.h, _p(buf), len(buf), mode, min_len)

    @staticmethod
    def from_arrays(c, k, marker_c, file_name, seed, pos, ctgcanon, markers, contig_lengths, total_len):
        seed = np.ascontiguousarray(seed, np.uint32); pos = np.ascontiguousarray(pos, np.uint32)
        cc = np.ascontiguousarray(ctgcanon, np.uint32); mk = np.ascontiguousarray(markers, np.uint64)
        cl = np.ascontiguousarray(contig_lengths, np.uint32)
        h = lib().ora_sketch_from_arrays(c, k, marker_c, file_name.encode(), _p(seed), _p(pos), _p(cc), len(seed),
                                         _p(mk), len(mk), _p(cl), len(cl), int(total_len))
        return Sketch(c, k, marker_c, file_name, handle=h)

    @property
    def n_positions(self): return lib().ora_sketch_n_positions(self.h)
    @property
    def n_distinct(self): return lib().ora_sketch_n_distinct(self.h)
    @property
    def n_markers(self): return lib().ora_sketch_n_markers(self.h)
    @property
    def n_contigs(self): return lib().ora_sketch_n_contigs(self.h)
    @property
    def total_len(self): return lib().ora_sketch_total_len(self.h)

    def seeds(self, pos_order=False):
        n = self.n_positions
        s = np.empty(n, np.uint32); p = np.empty(n, np.uint32); cc = np.empty(n, np.uint32)
        (lib().ora_sketch_export_seeds_pos_order if pos_order else lib().ora_sketch_export_seeds)(self.h, _p(s), _p(p), _p(cc))
        return s, p, cc

    def markers(self):
        m = np.empty(self.n_markers, np.uint64); lib().ora_sketch_export_markers(self.h, _p(m)); return m

    def contig_lengths(self):
        l = np.empty(self.n_contigs, np.uint32); lib().ora_sketch_export_contig_lengths(self.h, _p(l)); return l


class Model:
    def __init__(self, path):
        self.h = lib().ora_model_load(path.encode())
        if not self.h:
            raise IOError("cannot load GBDT table " + path)

    def predict(self, feat):
        f = np.ascontiguousarray(feat, np.float32); return lib().ora_model_predict(self.h, _p(f))


def sketch_records(records, c=125, k=15, marker_c=1000, file_name="", mode=1, min_len=500):
    """file_io.rs:141-252 for one file: records = iterable of (name, seq bytes)."""
    sk = Sketch(c, k, marker_c, file_name)
    for _, seq in records:
        sk.add_contig(seq, mode, min_len)
    return sk


def sketch_batch(genomes, c=125, k=15, marker_c=1000, names=None, mode=1, min_len=500, threads=0):
    """file_io.rs:141-252 for many files at once, sketched in parallel inside one C call (file_io.rs:147): genomes = list of lists of
    (name, seq) records, seq = bytes or a uint8 numpy array."""
    bufs = [np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else np.ascontiguousarray(s, np.uint8) for g in genomes for _, s in g]
    off = np.zeros(len(genomes) + 1, np.uint64); off[1:] = np.cumsum([len(g) for g in genomes])
    ptrs = (C.c_void_p * max(len(bufs), 1))(*[b.ctypes.data for b in bufs])
    lens = np.array([len(b) for b in bufs], np.uint64)
    nm = [(names[i] if names else "").encode() for i in range(len(genomes))]
    cn = (C.c_char_p * max(len(nm), 1))(*nm)
    out = (C.c_void_p * max(len(genomes), 1))()
    lib().ora_sketch_batch(len(genomes), _p(off), ptrs, _p(lens), c, k, marker_c, cn, mode, min_len, threads, out)
    return [Sketch(c, k, marker_c, names[i] if names else "", handle=out[i]) for i in range(len(genomes))]


def sketch_files(paths, c=125, k=15, marker_c=1000, mode=1, min_len=500, threads=0):
    """fastx_to_sketches (file_io.rs:141-252) on plain FASTA files: the sketches of the files that have a kept contig, in the order given."""
    arr = (C.c_char_p * max(len(paths), 1))(*[p.encode() for p in paths])
    out = (C.c_void_p * max(len(paths), 1))()
    lib().ora_sketch_files(len(paths), arr, c, k, marker_c, mode, min_len, threads, out)
    return [Sketch(c, k, marker_c, paths[i], handle=out[i]) for i in range(len(paths)) if out[i]]


def chain_seeds(ref, query, min_af=0.15, both_min_af=-0.01, robust=False, median=False, model=None, stats=False):
    mo = MapOpts(min_af, both_min_af, int(robust), int(median)); r = AniResult(); st = ChainStats()
    lib().ora_chain_seeds(ref.h, query.h, C.byref(mo), model.h if model else None, C.byref(r), C.byref(st))
    return (r, st) if stats else r


def check_markers_quickly(ref, query, screen_val, rescue_small):
    return bool(lib().ora_check_markers_quickly(ref.h, query.h, screen_val, int(rescue_small)))


def screen_refs(refs, query, identity=0.8, rule=0, rescue_small=True):
    arr = (C.c_void_p * len(refs))(*[r.h for r in refs]); out = np.empty(len(refs), np.uint32)
    n = lib().ora_screen_refs(arr, len(refs), query.h, identity, rule, int(rescue_small), _p(out))
    return out[:n].copy()


def triangle(sketches, screen_val=0.0, rescue_small=True, min_af=0.15, both_min_af=-0.01, robust=False, median=False,
             model=None, threads=0):
    n = len(sketches); arr = (C.c_void_p * n)(*[s.h for s in sketches])
    cap = n * (n - 1) // 2 + 1
    oi = np.empty(cap, np.uint32); oj = np.empty(cap, np.uint32); res = np.zeros(cap, RESULT_DTYPE)
    mo = MapOpts(min_af, both_min_af, int(robust), int(median)); nch = C.c_uint64(); nsp = C.c_uint64()
    kept = lib().ora_triangle(arr, n, screen_val, int(rescue_small), C.byref(mo), model.h if model else None, threads,
                              _p(oi), _p(oj), _p(res), cap, C.byref(nch), C.byref(nsp))
    return oi[:kept].copy(), oj[:kept].copy(), res[:kept].copy(), nch.value, nsp.value


def search(refs, queries, screen_val=0.0, use_index=True, min_af=0.15, both_min_af=-0.01, robust=False, median=False, model=None, threads=0, cap=None):
    """search.rs:97-200 with every reference resident: (query, ref, results, n_chained), ani > 0.5, in (query, ref) order; phases through triangle_phases()."""
    nr, nq = len(refs), len(queries)
    ra = (C.c_void_p * max(nr, 1))(*[s.h for s in refs]); qa = (C.c_void_p * max(nq, 1))(*[s.h for s in queries])
    cap = cap or (nr * nq + 1)
    oq = np.empty(cap, np.uint32); orf = np.empty(cap, np.uint32); res = np.zeros(cap, RESULT_DTYPE)
    mo = MapOpts(min_af, both_min_af, int(robust), int(median)); nch = C.c_uint64()
    kept = lib().ora_search(ra, nr, qa, nq, screen_val, int(use_index), C.byref(mo), model.h if model else None, threads, _p(oq), _p(orf), _p(res), cap, C.byref(nch))
    return oq[:kept].copy(), orf[:kept].copy(), res[:kept].copy(), nch.value


def triangle_phases():
    """(index_s, screen_s, chain_s) of the last triangle() call."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().ora_triangle_phases(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value
