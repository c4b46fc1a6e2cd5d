"""skani's on-disk formats (skani_amd/host/formats.cpp), CPU-only.

Pins: the reference's bundled pre-0.3 sketch (tests/golden/e.coli-o157.fasta.sketch, a data file of the reference's own
test-suite) must decode to the golden seed set; v0.3 files written by the C++ writer are re-read by an independent
byte-level decoder written here from the serde field order (types.rs:252-277, params.rs:136-146, sketch_db.rs:10-15)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from skani_amd.build import build_hip, build_host
from tests.helpers import GOLDEN


@pytest.fixture(scope="module")
def host():
    from tests.host_shims.build_shims import build as build_shims
    L = C.CDLL(build_shims())                                         # the host sources behind test-only extern "C" hooks (tests/host_shims)
    for f in ("skhost_sketch_read", "skhost_sketch_write", "skhost_db_write", "skhost_db_summary", "skhost_sketch_summary"):
        getattr(L, f).restype = C.c_void_p
    return L


def _take(L, p):
    if not p:
        return None
    s = C.string_at(p).decode(); L.skhost_test_free(C.c_void_p(p)); return s


def read_sketch(L, path):
    ckm = (C.c_uint64 * 3)(); fmt = C.c_int(); n_rec = C.c_uint64(); n_mk = C.c_uint64(); n_ctg = C.c_uint64()
    seed = C.POINTER(C.c_uint32)(); pos = C.POINTER(C.c_uint32)(); cc = C.POINTER(C.c_uint32)(); clen = C.POINTER(C.c_uint32)()
    mk = C.POINTER(C.c_uint64)(); scal = (C.c_uint64 * 6)(); names = C.c_void_p()
    err = _take(L, L.skhost_sketch_read(str(path).encode(), ckm, C.byref(fmt), C.byref(n_rec), C.byref(seed), C.byref(pos), C.byref(cc), C.byref(n_mk),
                                        C.byref(mk), C.byref(n_ctg), C.byref(clen), scal, C.byref(names)))
    if err:
        raise RuntimeError(err)
    def arr(p, n, dt):
        a = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt).copy() if n else np.zeros(0, dt)
        L.skhost_test_free(C.cast(p, C.c_void_p)); return a
    nm = C.string_at(names).decode().split("\n"); L.skhost_test_free(names)
    return dict(c=ckm[0], k=ckm[1], m=ckm[2], format=fmt.value, seed=arr(seed, n_rec.value, np.uint32), pos=arr(pos, n_rec.value, np.uint32),
                ctgcanon=arr(cc, n_rec.value, np.uint32), markers=arr(mk, n_mk.value, np.uint64), contig_lengths=arr(clen, n_ctg.value, np.uint32),
                total_len=scal[0], sk_marker_c=scal[1], sk_c=scal[2], sk_k=scal[3], contig_order=scal[4], repetitive=scal[5], file_name=nm[0], contigs=nm[1:])


def write_sketch(L, path, d):
    seed = np.ascontiguousarray(d["seed"], np.uint32); pos = np.ascontiguousarray(d["pos"], np.uint32); cc = np.ascontiguousarray(d["ctgcanon"], np.uint32)
    mk = np.ascontiguousarray(d["markers"], np.uint64); clen = np.ascontiguousarray(d["contig_lengths"], np.uint32)
    names = (C.c_char_p * len(d["contigs"]))(*[s.encode() for s in d["contigs"]])
    ckm = (C.c_uint64 * 3)(d["c"], d["k"], d["m"])
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    err = _take(L, L.skhost_sketch_write(str(path).encode(), ckm, d["file_name"].encode(), C.c_uint64(len(seed)), p(seed, C.c_uint32), p(pos, C.c_uint32),
                                         p(cc, C.c_uint32), C.c_uint64(len(mk)), p(mk, C.c_uint64), C.c_uint64(len(clen)), names, p(clen, C.c_uint32),
                                         C.c_uint64(int(d["total_len"])), C.c_uint64(int(d.get("contig_order", 0)))))
    if err:
        raise RuntimeError(err)


# ---- independent byte-level decoder of the v0.3 layout (bincode 1.x defaults) ----
class Rd:
    def __init__(self, b): self.b, self.at = b, 0
    def u8(self): v = self.b[self.at]; self.at += 1; return v
    def u32(self): v = struct.unpack_from("<I", self.b, self.at)[0]; self.at += 4; return v
    def u64(self): v = struct.unpack_from("<Q", self.b, self.at)[0]; self.at += 8; return v
    def s(self): n = self.u64(); v = self.b[self.at:self.at + n].decode(); self.at += n; return v


def py_params(r):
    p = dict(c=r.u64(), k=r.u64(), m=r.u64(), use_syncs=r.u8(), use_aa=r.u8())
    p["enc"] = [r.u64() for _ in range(r.u64())]
    n = r.u64(); p["letters"] = bytes(r.b[r.at:r.at + n]); r.at += n
    p["orf"] = r.u64()
    return p


def py_sketch(r):
    d = dict(file_name=r.s())
    recs = []
    tag = r.u8(); entries = []
    if tag:
        entries = [(r.u32(), r.u64()) for _ in range(r.u64())]
    multi = []
    for _ in range(r.u64()):
        multi.append([(r.u32(), r.u32()) for _ in range(r.u64())])
    for seed, tagged in entries:
        if tagged & 1:
            packed = tagged >> 1
            recs.append((seed, packed >> 31, packed & 0x7FFFFFFF))
        else:
            recs += [(seed, p, c) for p, c in multi[tagged >> 1]]
    d["has_seeds"] = bool(tag); d["n_multi"] = len(multi); d["multi"] = multi
    d["contigs"] = [r.s() for _ in range(r.u64())]
    d["total_len"] = r.u64()
    d["contig_lengths"] = [r.u32() for _ in range(r.u64())]
    d["repetitive"] = r.u64()
    d["markers"] = sorted(r.u64() for _ in range(r.u64()))
    d["sk_marker_c"], d["sk_c"], d["sk_k"], d["contig_order"] = r.u64(), r.u64(), r.u64(), r.u64()
    d["individual_contig"], d["amino_acid"] = r.u8(), r.u8()
    recs.sort(key=lambda t: (t[2] >> 1, t[1]))
    d["records"] = recs
    return d


def golden_o157():
    z = np.load(os.path.join(GOLDEN, "o157_sketch.npz"))
    o = np.lexsort((z["pos"], z["ctgcanon"] >> 1))
    return z, o


def test_legacy_golden_sketch_decodes_to_the_golden_seed_set(host):
    d = read_sketch(host, os.path.join(GOLDEN, "e.coli-o157.fasta.sketch"))
    z, o = golden_o157()
    assert d["format"] == 2 and (d["c"], d["k"], d["m"]) == (125, 15, 1000)
    assert len(d["seed"]) == 44127 and len(np.unique(d["seed"])) == 40716 and len(d["markers"]) == 5073      # SURVEY 8c pin 1
    assert np.array_equal(d["seed"], z["seed"][o]) and np.array_equal(d["pos"], z["pos"][o]) and np.array_equal(d["ctgcanon"], z["ctgcanon"][o])
    assert np.array_equal(d["markers"], np.sort(z["markers"]))
    assert list(d["contig_lengths"]) == [5416633, 92596] and d["total_len"] == 5509229
    assert d["file_name"] == "test_files/e.coli-o157.fasta" and d["contigs"] == list(z["contigs"])


def test_v03_writer_layout_and_round_trip(host, tmp_path):
    src = read_sketch(host, os.path.join(GOLDEN, "e.coli-o157.fasta.sketch"))
    src["contig_order"] = 0
    p = tmp_path / "o157.sketch"
    write_sketch(host, p, src)
    back = read_sketch(host, p)
    assert back["format"] == 3
    for k in ("seed", "pos", "ctgcanon", "markers", "contig_lengths"):
        assert np.array_equal(back[k], src[k]), k
    for k in ("c", "k", "m", "total_len", "file_name", "contigs"):
        assert back[k] == src[k], k
    # byte layout, decoded independently
    r = Rd(p.read_bytes()); pp = py_params(r); d = py_sketch(r)
    assert r.at == len(r.b)
    assert (pp["c"], pp["k"], pp["m"], pp["use_syncs"], pp["use_aa"], pp["orf"]) == (125, 15, 1000, 0, 0, 30)
    # codon table in A,C,G,T order and the amino-acid numbering of params.rs:151-174 (R = 15: the later duplicate wins)
    assert pp["letters"] == b"KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"
    order = {c: i for i, c in enumerate("ARNDCEFGHIKLMPQRSTVWY")}; order["*"] = 21
    assert pp["enc"] == [order[chr(c)] for c in pp["letters"]] and order["R"] == 15
    assert d["records"] == list(zip(src["seed"].tolist(), src["pos"].tolist(), src["ctgcanon"].tolist()))
    assert d["markers"] == src["markers"].tolist() and d["contig_lengths"] == [5416633, 92596] and d["total_len"] == 5509229
    assert (d["sk_marker_c"], d["sk_c"], d["sk_k"], d["individual_contig"], d["amino_acid"], d["repetitive"]) == (125, 125, 15, 0, 0, 0)   # types.rs:342-351
    # multi_position_storage holds exactly the seeds with more than one position, each list in (contig, pos) order
    seeds, counts = np.unique(src["seed"], return_counts=True)
    assert d["n_multi"] == int((counts > 1).sum()) and sorted(len(m) for m in d["multi"]) == sorted(counts[counts > 1].tolist())
    assert all(m == sorted(m, key=lambda t: (t[1] >> 1, t[0])) for m in d["multi"])


def test_corrupt_and_truncated_files_are_rejected(host, tmp_path):
    good = open(os.path.join(GOLDEN, "e.coli-o157.fasta.sketch"), "rb").read()
    for name, data in (("empty", b""), ("short", good[:1000]), ("cut", good[:-3]), ("extra", good + b"\0"), ("text", b">seq\nACGT\n" * 100),
                       ("hugelen", good[:24] + b"\xff" * 8 + good[32:])):
        p = tmp_path / (name + ".sketch"); p.write_bytes(data)
        with pytest.raises(RuntimeError):
            read_sketch(host, p)
        assert _take(host, host.skhost_sketch_summary(str(p).encode())).startswith("ERROR")


@pytest.mark.parametrize("separate", [0, 1])
def test_database_folder_round_trip(host, tmp_path, separate):
    src = read_sketch(host, os.path.join(GOLDEN, "e.coli-o157.fasta.sketch"))
    files = []
    for i, (name, keep) in enumerate((("refs/b_chrom.fa", 0), ("refs/a_plasmid.fa", 1))):     # split the two contigs into two sketches
        sel = (src["ctgcanon"] >> 1) == keep
        d = dict(src); d["file_name"] = name
        d["seed"], d["pos"], d["ctgcanon"] = src["seed"][sel], src["pos"][sel], src["ctgcanon"][sel] & 1
        d["contig_lengths"] = src["contig_lengths"][keep:keep + 1]; d["contigs"] = src["contigs"][keep:keep + 1]; d["total_len"] = int(src["contig_lengths"][keep])
        d["markers"] = src["markers"][i::2]
        p = tmp_path / ("in%d.sketch" % i); write_sketch(host, p, d); files.append(str(p))
    db = tmp_path / "db"; db.mkdir()
    arr = (C.c_char_p * 2)(*[f.encode() for f in files])
    assert _take(host, host.skhost_db_write(str(db).encode(), 2, arr, separate)) is None
    want = sorted(["markers.bin", "index.db", "sketches.db"] if not separate else ["markers.bin", "a_plasmid.fa.sketch", "b_chrom.fa.sketch"])
    assert sorted(os.listdir(db)) == want
    summ = _take(host, host.skhost_db_summary(str(db).encode())).split("\n")
    assert summ[0] == "params\t125\t15\t1000"
    full, mk = summ[1:3], summ[4:6]
    assert summ[3] == "markers"
    assert [l.split("\t")[0] for l in full] == ["refs/a_plasmid.fa", "refs/b_chrom.fa"]              # sorted by file name (file_io.rs:727)
    a = full[0].split("\t"); assert a[1:5] == ["759", str(len(src["markers"][1::2])), "1", "92596"]   # 759 plasmid seeds: SURVEY 8c pin 1
    for f, m in zip(full, mk):                                                                        # marker sketches: no seeds, no contig lengths
        f, m = f.split("\t"), m.split("\t")
        assert m[0] == f[0] and m[1] == "0" and m[2] == f[2] and m[4] == f[4]
    # same sketches as the single files
    assert full[0] == _take(host, host.skhost_sketch_summary(files[1].encode())).strip()
    # independent decode of markers.bin (+ index.db / sketches.db)
    r = Rd((db / "markers.bin").read_bytes()); pp = py_params(r); n = r.u64(); ms = [py_sketch(r) for _ in range(n)]
    assert r.at == len(r.b) and n == 2 and [m["has_seeds"] for m in ms] == [False, False] and [m["contig_lengths"] for m in ms] == [[], []]
    assert ms[0]["markers"] == src["markers"][1::2].tolist() and ms[0]["total_len"] == 92596 and ms[0]["contigs"] == src["contigs"][1:2]
    if not separate:
        r = Rd((db / "index.db").read_bytes()); idx = [(r.s(), r.u64(), r.u64()) for _ in range(r.u64())]
        assert r.at == len(r.b) and [e[0] for e in idx] == ["refs/a_plasmid.fa", "refs/b_chrom.fa"] and idx[0][1] == 0 and idx[1][1] == idx[0][2]
        blob = (db / "sketches.db").read_bytes(); assert len(blob) == idx[1][1] + idx[1][2]
        r = Rd(blob[idx[1][1]:]); py_params(r); d = py_sketch(r)
        assert d["file_name"] == "refs/b_chrom.fa" and len(d["records"]) == 44127 - 759
