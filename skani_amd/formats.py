"""skani's on-disk sketch formats above the C ABI (host/formats.cpp through libskani_host.so).

`load_database` / `load_sketch_files` put a folder written by `skani sketch` (sketches.db + index.db + markers.bin, or separate
.sketch files) or loose .sketch files into HBM as one SketchSet -- sketches_from_sketch (file_io.rs:680-729) and the database
reader of search.rs:16-100 without the lazy fetch.  `save_database` is `skani sketch -o dir` (sketch.rs:15-175) for a resident set."""
import ctypes as C
import os

import numpy as np

from . import _binding as B
from .build import build_host

_HOST = None


def _host():
    global _HOST
    if _HOST is None:
        lib, _ = build_host()
        L = C.CDLL(lib)
        for f in ("skhost_db_open", "skhost_db_save"):
            getattr(L, f).restype = C.c_void_p
        L.skhost_db_dims.restype = None; L.skhost_db_fill.restype = None; L.skhost_db_close.restype = None
        _HOST = L
    return _HOST


def _err(L, p):
    if not p:
        return None
    s = C.string_at(p).decode(); L.skhost_free(C.c_void_p(p)); return s


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _load(ctx, path, kind, seeding_mode):
    L = _host(); h = C.c_void_p()
    e = _err(L, L.skhost_db_open(path.encode(), kind, C.byref(h)))
    if e:
        from .api import SkaniHipError
        raise SkaniHipError(e)
    try:
        dims = (C.c_uint64 * 8)(); L.skhost_db_dims(h, dims)
        n, P, M, NC, c, k, m, nb = [int(x) for x in dims]
        po = np.zeros(n + 1, np.uint64); mo = np.zeros(n + 1, np.uint64); co = np.zeros(n + 1, np.uint64)
        seed = np.zeros(P, np.uint32); pos = np.zeros(P, np.uint32); cc = np.zeros(P, np.uint32); mk = np.zeros(M, np.uint64)
        cl = np.zeros(NC, np.uint32); tl = np.zeros(n, np.uint64); order = np.zeros(n, np.uint64); names = C.create_string_buffer(nb + 1)
        L.skhost_db_fill(h, _p(po), _p(seed), _p(pos), _p(cc), _p(mo), _p(mk), _p(co), _p(cl), _p(tl), _p(order), names)
    finally:
        L.skhost_db_close(h)
    lines = names.raw[:nb].decode().split("\n")
    infos, at = [], 0
    for g in range(n):
        nc = int(co[g + 1] - co[g])
        infos.append(dict(file_name=lines[at], contigs=lines[at + 1:at + 1 + nc], contig_order=int(order[g]))); at += 1 + nc
    params = B.SketchParams(c, k, m, seeding_mode)
    file_names = [i["file_name"] for i in infos]
    o = sorted(range(n), key=lambda i: (file_names[i], infos[i]["contig_order"])); rank = np.empty(n, np.uint32); rank[o] = np.arange(n, dtype=np.uint32)
    meta = dict(pos_off=po, marker_off=mo, contig_off=co, contig_lengths=cl, total_len=tl, genome_rank=rank)
    ss = ctx.import_flat(params, meta, seed, pos, cc, mk, device=False, names=file_names)
    return ss, infos


def load_database(ctx, folder, seeding_mode=1):
    """-> (SketchSet in index order, [dict(file_name, contigs, contig_order)])."""
    return _load(ctx, str(folder), 0, seeding_mode)


def load_sketch_files(ctx, files, seeding_mode=1):
    """.sketch files (v0.3 layout, or the pre-0.3 layout of the reference's bundled test sketch), sorted by file name."""
    return _load(ctx, "\n".join(str(f) for f in files), 1, seeding_mode)


def save_database(ss, folder, infos=None, separate_files=False, individual_contig=False):
    """Write a resident SketchSet as a skani database folder (which must not exist, sketch.rs:19-23)."""
    L = _host()
    if os.path.exists(folder):
        raise FileExistsError("Output directory exists; output directory must not be an existing directory.")
    os.makedirs(folder)
    n = len(ss); P, M, NC = ss.totals(); meta = ss.export_meta()
    seed = np.zeros(P, np.uint32); pos = np.zeros(P, np.uint32); cc = np.zeros(P, np.uint32); mk = np.zeros(M, np.uint64)
    ss.export_arrays(seed, pos, cc, mk)
    if infos is None:
        names = ss.names if ss.names is not None else ["genome%d" % g for g in range(n)]
        infos = [dict(file_name=names[g], contigs=["contig%d" % c for c in range(int(meta["contig_off"][g + 1] - meta["contig_off"][g]))], contig_order=0)
                 for g in range(n)]
    blob = "".join(i["file_name"] + "\n" + "".join(c + "\n" for c in i["contigs"]) for i in infos).encode()
    order = np.array([i.get("contig_order", 0) for i in infos], np.uint64)
    ckm = (C.c_uint64 * 3)(ss.params.c, ss.params.k, ss.params.marker_c)
    e = _err(L, L.skhost_db_save(str(folder).encode(), int(separate_files), int(individual_contig), ckm, C.c_uint64(n), _p(meta["pos_off"]), _p(seed), _p(pos), _p(cc),
                                 _p(meta["marker_off"]), _p(mk), _p(meta["contig_off"]), _p(meta["contig_lengths"]), _p(meta["total_len"]), _p(order), blob))
    if e:
        raise RuntimeError(e)
