"""Host-side mirror of the reference's operator API for the hot path (names follow src/lib.rs:1-22):
fastx_to_sketches (file_io.rs:141), screen_refs / check_markers_quickly (screen.rs), map_params_from_sketch +
chain_seeds (chain.rs:88,144), the triangle body (triangle.rs:55-105) -- all at batch granularity."""
import ctypes as C
import os

import numpy as np

from . import _binding as B
from .fastx import read_fasta

SEED_SCALAR, SEED_AVX2 = 0, 1
MIN_LENGTH_CONTIG = 500          # params.rs:42
_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = None


class SkaniHipError(RuntimeError):
    pass


def SketchParams(c=125, k=15, marker_c=1000, seeding_mode=SEED_AVX2):
    """reference SketchParams::new (params.rs:148-196); seeding_mode selects which reference seeding path's
    semantics are reproduced (x86-64 hosts run the AVX2 one, file_io.rs:194-206)."""
    return B.SketchParams(c, k, marker_c, seeding_mode)


def MapParams(min_af=0.15, both_min_af=-0.01, robust=False, median=False, learned_ani=False, compute_ci=False):
    return B.MapParams(min_af, both_min_af, int(robust), int(median), int(learned_ani), int(compute_ci))


def use_learned_ani(c, individual_contig_q=False, individual_contig_r=False, median=False):
    """regression.rs:8-10"""
    return c >= 70 and not individual_contig_q and not individual_contig_r and not median


def _default_lib():
    global _DEFAULT_LIB
    if _DEFAULT_LIB is None:
        path = os.path.join(_HERE, "libskani_hip.so")
        if not os.path.exists(path):
            raise SkaniHipError(f"{path} is missing: build it with `python -m skani_amd.build` (hipcc, gfx950). "
                                "skani_amd has no CPU fallback.")
        # PyTorch-ROCm bundles its own HIP runtime under the same soname (libamdhip64.so.7) as /opt/rocm's: whichever is
        # loaded first serves the whole process.  Torch cannot run on the system runtime, the library can run on
        # torch's, so torch (when installed) goes first.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _DEFAULT_LIB = B.load(path)
    return _DEFAULT_LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One GPU.  `lib` is only overridden by the kernel-simulator tests (tests/emu)."""

    def __init__(self, device=0, lib=None, load_models=True):
        self.L = lib if lib is not None else _default_lib()
        h = C.c_void_p()
        rc = self.L.skh_ctx_create(device, C.byref(h))
        if rc != 0:
            raise SkaniHipError(f"skh_ctx_create(device={device}) failed with {rc}: no usable gfx950 device (no CPU path)")
        self.h = h
        if load_models:
            self.check(self.L.skh_load_models(self.h, os.path.join(_HERE, "data", "gbdt_c125.bin").encode(),
                                              os.path.join(_HERE, "data", "gbdt_c200.bin").encode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.skh_ctx_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise SkaniHipError(f"skani_hip error {rc}: {self.L.skh_last_error(self.h).decode()}")

    def timings(self):
        t = B.Timings(); self.check(self.L.skh_get_timings(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_ if n != "pad"}

    def device_memory(self, trim=False):
        """(live, idle) bytes of device memory the library holds: in use / freed blocks kept by its caching allocator (skh_device_memory)."""
        live, idle = C.c_uint64(), C.c_uint64()
        self.check(self.L.skh_device_memory(C.byref(live), C.byref(idle), int(trim)))
        return live.value, idle.value

    # ---- ingest -------------------------------------------------------------------------------------------------
    def pack_genomes(self, genomes, seeding_mode=SEED_AVX2):
        """genomes: list (one per genome) of lists of contig byte strings (already >= 500 bp filtered)."""
        contig_genome, lens = [], []
        for g, ctgs in enumerate(genomes):
            for s in ctgs:
                contig_genome.append(g); lens.append(len(s))
        off = np.zeros(len(lens) + 1, np.uint64); off[1:] = np.cumsum(np.array(lens, np.uint64))
        bases = np.frombuffer(b"".join(s for ctgs in genomes for s in ctgs), np.uint8) if lens else np.zeros(1, np.uint8)
        return self.pack_buffer(bases, off, np.array(contig_genome, np.uint32), len(genomes), seeding_mode)

    def pack_buffer(self, bases, contig_off, contig_genome, n_genomes, seeding_mode=SEED_AVX2, device_ptr=None):
        """bases: numpy uint8 (host) or, with device_ptr, a raw device address of ASCII bases already in HBM."""
        contig_off = np.ascontiguousarray(contig_off, np.uint64); contig_genome = np.ascontiguousarray(contig_genome, np.uint32)
        h = C.c_void_p()
        ptr = C.c_void_p(device_ptr) if device_ptr is not None else _p(np.ascontiguousarray(bases, np.uint8))
        self.check(self.L.skh_genomes_pack(self.h, ptr, _p(contig_off), _p(contig_genome), len(contig_genome), n_genomes,
                                           1 if device_ptr is not None else 0, seeding_mode, C.byref(h)))
        return GenomeSet(self, h, seeding_mode, contig_off, contig_genome, n_genomes)

    def pack_batches(self, batches, seeding_mode=SEED_AVX2, max_bases=None, max_contigs=None):
        """skh_genomes_begin / _append / _finish: `batches` = list of batches, a batch = list of (genome number, [contig byte strings]); the genome
        numbers of all batches together are 0 .. n-1 in any order.  Every batch goes through its own pinned buffer with a gap between the contigs
        (what parser threads writing files into stretches of one buffer produce)."""
        n_ctg = sum(len(c) for b in batches for _, c in b); n_bases = sum(len(s) for b in batches for _, c in b for s in c)
        n_genomes = 1 + max([g for b in batches for g, _ in b], default=-1)
        h = C.c_void_p()
        self.check(self.L.skh_genomes_begin(self.h, max_bases if max_bases is not None else n_bases, max_contigs if max_contigs is not None else n_ctg, n_genomes,
                                            seeding_mode, C.byref(h)))
        pins = []
        try:
            for b in batches:
                starts, lens, cg, at = [], [], [], 3
                for g, ctgs in b:
                    for s in ctgs:
                        starts.append(at); lens.append(len(s)); cg.append(g); at += len(s) + 5      # five stray bytes between contigs
                pin = self.L.skh_host_alloc(at + 8)
                if not pin:
                    raise SkaniHipError("skh_host_alloc failed")
                pins.append(pin)
                view = np.ctypeslib.as_array((C.c_uint8 * (at + 8)).from_address(pin)); view[:] = ord("#")
                k = 0
                for _, ctgs in b:
                    for s in ctgs:
                        view[starts[k]:starts[k] + lens[k]] = np.frombuffer(s, np.uint8); k += 1
                st = np.array(starts, np.uint64); ln = np.array(lens, np.uint64); cgn = np.array(cg, np.uint32)
                ticket = C.c_uint64()
                self.check(self.L.skh_genomes_append(h, C.c_void_p(pin), _p(st), _p(ln), _p(cgn), len(starts), 0, C.byref(ticket)))
                self.check(self.L.skh_genomes_wait(h, ticket.value))
            self.check(self.L.skh_genomes_finish(h))
        except Exception:
            self.L.skh_genomes_destroy(h)
            raise
        finally:
            for pin in pins:
                self.L.skh_host_free(pin)
        return GenomeSet(self, h, seeding_mode, None, None, n_genomes)

    # ---- sketch -------------------------------------------------------------------------------------------------
    def sketch_genomes(self, gs, params, genome_rank=None, names=None, defer_tables=False, screen_index=True, compact=False):
        """defer_tables: seeding + marker sets only; the seed tables are built on first use (a rank of a distributed triangle indexes only what it chains;
        `triangle` builds them beside its screen).  screen_index=False (with defer_tables): no sorted marker incidences either (SKH_SKETCH_NO_SCREEN_INDEX)."""
        h = C.c_void_p()
        rank = np.ascontiguousarray(genome_rank, np.uint32) if genome_rank is not None else None
        flags = (1 if defer_tables else 0) | (0 if screen_index else 2) | (4 if compact else 0)   # compact: SKH_SKETCH_COMPACT, a set that stays resident (search database)
        self.check(self.L.skh_sketch_genomes_ex(self.h, gs.h, C.byref(params), _p(rank), flags, C.byref(h)))
        return SketchSet(self, h, params, names)

    def sketch_records(self, genomes, params, names=None, defer_tables=False, screen_index=True, compact=False):
        """genomes: list of lists of (name, seq) records for one file each; applies file_io.rs:176 (>= 500 bp)."""
        kept = [[s for _, s in recs if len(s) >= MIN_LENGTH_CONTIG] for recs in genomes]
        rank = None
        if names is not None:
            order = sorted(range(len(names)), key=lambda i: names[i]); rank = np.empty(len(names), np.uint32)
            rank[order] = np.arange(len(names), dtype=np.uint32)
        gs = self.pack_genomes(kept, params.seeding_mode)
        try:
            return self.sketch_genomes(gs, params, rank, names, defer_tables=defer_tables, screen_index=screen_index, compact=compact)
        finally:
            gs.close()

    def import_sketches(self, params, per_genome, names=None, genome_rank=None):
        """per_genome: list of dicts with seed,pos,ctgcanon (position order), markers (sorted), contig_lengths, total_len."""
        ng = len(per_genome)
        def cat(key, dt):
            return np.ascontiguousarray(np.concatenate([np.asarray(d[key], dt) for d in per_genome]) if ng else np.zeros(0, dt), dt)
        def offs(key):
            o = np.zeros(ng + 1, np.uint64); o[1:] = np.cumsum([len(d[key]) for d in per_genome]); return o
        order = np.lexsort  # noqa
        fixed = []
        for d in per_genome:   # make sure seeds are in (contig, pos) order
            cc = np.asarray(d["ctgcanon"], np.uint32); ps = np.asarray(d["pos"], np.uint32)
            o = np.lexsort((ps, cc >> 1))
            fixed.append(dict(seed=np.asarray(d["seed"], np.uint32)[o], pos=ps[o], ctgcanon=cc[o], markers=np.sort(np.asarray(d["markers"], np.uint64)),
                              contig_lengths=np.asarray(d["contig_lengths"], np.uint32), total_len=int(d["total_len"])))
        per_genome = fixed
        seed, pos, cc = cat("seed", np.uint32), cat("pos", np.uint32), cat("ctgcanon", np.uint32)
        mk, cl = cat("markers", np.uint64), cat("contig_lengths", np.uint32)
        tl = np.array([d["total_len"] for d in per_genome], np.uint64)
        rank = np.ascontiguousarray(genome_rank, np.uint32) if genome_rank is not None else None
        if rank is None and names is not None:
            o = sorted(range(ng), key=lambda i: names[i]); rank = np.empty(ng, np.uint32); rank[o] = np.arange(ng, dtype=np.uint32)
        h = C.c_void_p()
        po, mo, co = offs("seed"), offs("markers"), offs("contig_lengths")
        self.check(self.L.skh_sketch_import(self.h, C.byref(params), ng, _p(po), _p(seed), _p(pos), _p(cc), _p(mo), _p(mk), _p(co), _p(cl),
                                            _p(tl), _p(rank), C.byref(h)))
        return SketchSet(self, h, params, names)

    def import_flat(self, params, meta, seed=None, pos=None, ctgcanon=None, markers=None, device=False, names=None):
        """meta: dict of host numpy arrays pos_off, marker_off, contig_off (n+1, uint64), contig_lengths (uint32), total_len (uint64),
        genome_rank (uint32).  The big arrays are numpy arrays (device=False) or raw device addresses (device=True)."""
        ng = len(meta["total_len"])
        po = np.ascontiguousarray(meta["pos_off"], np.uint64); mo = np.ascontiguousarray(meta["marker_off"], np.uint64)
        co = np.ascontiguousarray(meta["contig_off"], np.uint64); cl = np.ascontiguousarray(meta["contig_lengths"], np.uint32)
        tl = np.ascontiguousarray(meta["total_len"], np.uint64); rk = np.ascontiguousarray(meta["genome_rank"], np.uint32)
        def ptr(a, dt):
            if a is None:
                return None
            return C.c_void_p(int(a)) if device else _p(np.ascontiguousarray(a, dt))
        keep = [np.ascontiguousarray(a, dt) if (a is not None and not device) else a for a, dt in ((seed, np.uint32), (pos, np.uint32), (ctgcanon, np.uint32), (markers, np.uint64))]
        ptrs = [(C.c_void_p(int(a)) if device else _p(a)) if a is not None else None for a in keep]
        h = C.c_void_p()
        self.check(self.L.skh_sketch_import_flat(self.h, C.byref(params), ng, 1 if device else 0, _p(po), ptrs[0], ptrs[1], ptrs[2], _p(mo), ptrs[3], _p(co), _p(cl),
                                                 _p(tl), _p(rk), C.byref(h)))
        return SketchSet(self, h, params, names)

    # ---- screen / chain / triangle ------------------------------------------------------------------------------
    def screen(self, refs, queries=None, identity=0.0, rule=0, rescue_small=True):
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self.check(self.L.skh_screen(self.h, refs.h, queries.h if queries is not None else None, identity, rule, int(rescue_small),
                                     C.byref(a), C.byref(b), C.byref(n)))
        try:
            first = np.empty(n.value, np.uint32); second = np.empty(n.value, np.uint32)
            if n.value:
                C.memmove(first.ctypes.data, a, 4 * n.value); C.memmove(second.ctypes.data, b, 4 * n.value)
        finally:
            self.L.skh_free(a); self.L.skh_free(b)
        return first, second

    def screen_rows(self, sketches, row0, n_rows, identity=0.0, rescue_small=True):
        """The triangle's screen for rows [row0, row0 + n_rows): pairs (i, j), j > i (one GPU's share of a distributed triangle)."""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self.check(self.L.skh_screen_rows(self.h, sketches.h, row0, n_rows, identity, int(rescue_small), C.byref(a), C.byref(b), C.byref(n)))
        try:
            first = np.empty(n.value, np.uint32); second = np.empty(n.value, np.uint32)
            if n.value:
                C.memmove(first.ctypes.data, a, 4 * n.value); C.memmove(second.ctypes.data, b, 4 * n.value)
        finally:
            self.L.skh_free(a); self.L.skh_free(b)
        return first, second

    def screen_part(self, sketches, part, n_parts):
        """The triangle's screen cut by key range: the non-zero cells over part `part` of `n_parts` of the markers' leading 16 bases, one uint64 each
        (i << 43 | j << 22 | shared markers; `unpack_cells`)."""
        a, n = C.c_void_p(), C.c_uint64()
        self.check(self.L.skh_screen_part(self.h, sketches.h, part, n_parts, C.byref(a), C.byref(n)))
        try:
            cells = np.empty(n.value, np.uint64)
            if n.value: C.memmove(cells.ctypes.data, a, 8 * n.value)
        finally:
            self.L.skh_free(a)
        return cells

    @staticmethod
    def unpack_cells(cells):
        c = np.asarray(cells, np.uint64)
        return (c >> np.uint64(43)).astype(np.uint32), ((c >> np.uint64(22)) & np.uint64(0x1FFFFF)).astype(np.uint32), (c & np.uint64(0x3FFFFF)).astype(np.uint32)

    def screen_from_cells(self, sketches, cells, identity=0.0, rescue_small=True):
        """Candidate pairs (i, j > i) of the triangle from the concatenated cells of all key-range parts (skh_screen_from_cells)."""
        c = np.ascontiguousarray(cells, np.uint64)
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self.check(self.L.skh_screen_from_cells(self.h, sketches.h, _p(c), len(c), identity, int(rescue_small), C.byref(a), C.byref(b), C.byref(n)))
        try:
            first = np.empty(n.value, np.uint32); second = np.empty(n.value, np.uint32)
            if n.value:
                C.memmove(first.ctypes.data, a, 4 * n.value); C.memmove(second.ctypes.data, b, 4 * n.value)
        finally:
            self.L.skh_free(a); self.L.skh_free(b)
        return first, second

    def chain_pairs(self, refs, queries, pair_ref, pair_query, map_params, stats=False):
        """chain_seeds(refs[pair_ref[p]], queries[pair_query[p]], map_params_from_sketch(ref)) for every p (chain.rs:144)."""
        pr = np.ascontiguousarray(pair_ref, np.uint32); pq = np.ascontiguousarray(pair_query, np.uint32)
        out = np.zeros(len(pr), B.RESULT_DTYPE); st = np.zeros(len(pr), B.STATS_DTYPE) if stats else None
        self.check(self.L.skh_chain_pairs(self.h, refs.h, queries.h if queries is not None else None, _p(pr), _p(pq), len(pr),
                                          C.byref(map_params), _p(out), _p(st)))
        return (out, st) if stats else out

    def chain_pairs_multi(self, ref_sets, queries, pair_set, pair_ref, pair_query, map_params):
        """chain_seeds(ref_sets[pair_set[p]][pair_ref[p]], queries[pair_query[p]]) for every p: hits from all shards of a database in one call."""
        ps = np.ascontiguousarray(pair_set, np.uint32); pr = np.ascontiguousarray(pair_ref, np.uint32); pq = np.ascontiguousarray(pair_query, np.uint32)
        out = np.zeros(len(pr), B.RESULT_DTYPE)
        arr = (C.c_void_p * len(ref_sets))(*[s.h for s in ref_sets])
        self.check(self.L.skh_chain_pairs_multi(self.h, arr, len(ref_sets), queries.h, _p(ps), _p(pr), _p(pq), len(pr), C.byref(map_params), _p(out)))
        return out

    def triangle(self, sketches, map_params, identity=0.0, rescue_small=True, part=0, n_parts=1):
        oi, oj, orr, n, nch = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self.check(self.L.skh_triangle(self.h, sketches.h, identity, int(rescue_small), C.byref(map_params), part, n_parts,
                                       C.byref(oi), C.byref(oj), C.byref(orr), C.byref(n), C.byref(nch)))
        try:
            i, j, res = _take_rows(n.value, oi, oj, orr)
        finally:
            self.L.skh_free(oi); self.L.skh_free(oj); self.L.skh_free(orr)
        return i, j, res, nch.value


def _take_rows(k, oi, oj, orr):
    """Copies k result rows out of the library's malloc'd arrays (one memmove each: np.ctypeslib.as_array on a ctypes pointer costs ~0.1 ms per call)."""
    i = np.empty(k, np.uint32); j = np.empty(k, np.uint32); res = np.empty(k, B.RESULT_DTYPE)
    if k:
        C.memmove(i.ctypes.data, oi, 4 * k); C.memmove(j.ctypes.data, oj, 4 * k); C.memmove(res.ctypes.data, orr, B.RESULT_DTYPE.itemsize * k)
    return i, j, res


class GenomeSet:
    def __init__(self, ctx, h, mode, contig_off, contig_genome, n_genomes):
        self.ctx, self.h, self.mode = ctx, h, mode
        self.contig_off, self.contig_genome, self.n_genomes = contig_off, contig_genome, n_genomes

    @property
    def total_bases(self):
        return self.ctx.L.skh_genomes_total_bases(self.h)

    def close(self):
        if self.h:
            self.ctx.L.skh_genomes_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SketchSet:
    """Vec<Sketch> resident on the GPU."""

    def __init__(self, ctx, h, params, names=None):
        self.ctx, self.h, self.params, self.names = ctx, h, params, names
        if names is not None and len(names) == len(self):
            arr = (C.c_char_p * len(names))(*[str(n).encode() for n in names])
            ctx.check(ctx.L.skh_sketch_set_names(h, arr))

    def __len__(self):
        return self.ctx.L.skh_sketch_n_genomes(self.h)

    @property
    def wide(self):
        """The set holds a genome beyond 31-bit padded coordinates and keeps 64-bit ones (include/skani_hip.h skh_sketch_is_wide)."""
        return bool(self.ctx.L.skh_sketch_is_wide(self.h))

    def sizes(self, g):
        a, b, c_, d, e = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64()
        self.ctx.check(self.ctx.L.skh_sketch_sizes(self.h, g, C.byref(a), C.byref(b), C.byref(c_), C.byref(d), C.byref(e)))
        return dict(n_pos=a.value, n_distinct=b.value, n_markers=c_.value, n_contigs=d.value, total_len=e.value)

    def totals(self):
        a, b, c_ = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.ctx.check(self.ctx.L.skh_sketch_totals(self.h, C.byref(a), C.byref(b), C.byref(c_)))
        return a.value, b.value, c_.value

    def export_meta(self):
        """Per-genome host tables of the whole set (offsets, contig lengths, total lengths, ranks)."""
        n = len(self); P, M, NC = self.totals()
        meta = dict(pos_off=np.zeros(n + 1, np.uint64), marker_off=np.zeros(n + 1, np.uint64), contig_off=np.zeros(n + 1, np.uint64),
                    contig_lengths=np.zeros(NC, np.uint32), total_len=np.zeros(n, np.uint64), genome_rank=np.zeros(n, np.uint32))
        self.ctx.check(self.ctx.L.skh_sketch_export_flat(self.h, 0, None, None, None, None, _p(meta["pos_off"]), _p(meta["marker_off"]), _p(meta["contig_off"]),
                                                         _p(meta["contig_lengths"]), _p(meta["total_len"]), _p(meta["genome_rank"])))
        return meta

    def export_arrays(self, seed=None, pos=None, ctgcanon=None, markers=None, device=False):
        """Copy the set's big arrays into caller buffers: numpy arrays, or raw device addresses when device=True."""
        conv = (lambda a: C.c_void_p(int(a)) if a is not None else None) if device else (lambda a: _p(a) if a is not None else None)
        self.ctx.check(self.ctx.L.skh_sketch_export_flat(self.h, 1 if device else 0, conv(seed), conv(pos), conv(ctgcanon), conv(markers), None, None, None, None, None, None))

    def export(self, g):
        s = self.sizes(g)
        seed = np.empty(s["n_pos"], np.uint32); pos = np.empty(s["n_pos"], np.uint32); cc = np.empty(s["n_pos"], np.uint32)
        mk = np.empty(s["n_markers"], np.uint64); cl = np.empty(s["n_contigs"], np.uint32)
        self.ctx.check(self.ctx.L.skh_sketch_export(self.h, g, _p(seed), _p(pos), _p(cc), _p(mk), _p(cl)))
        return dict(seed=seed, pos=pos, ctgcanon=cc, markers=mk, contig_lengths=cl, total_len=s["total_len"])

    def close(self):
        if self.h:
            self.ctx.L.skh_sketch_set_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fastx_to_multiple_sketch_rewrite(ctx, files, params):
    """file_io.rs:253-362 (`-i`): one sketch per kept contig, ordered by (file name, contig order)."""
    genomes, names = [], []
    for f in sorted(files):
        for n, s in read_fasta(f):
            if len(s) >= MIN_LENGTH_CONTIG:
                genomes.append([(n, s)]); names.append(f)
    # the file name of every per-contig sketch goes into the set (skh_sketch_set_names): contigs of ONE file tie on it in switch_qr (chain.rs:20-22,
    # query_file_name > ref_file_name is false for equal names); rank = position in the (file, contig) order for sets without names
    ss = ctx.sketch_records(genomes, params, None)
    arr = (C.c_char_p * len(names))(*[str(n).encode() for n in names])
    ctx.check(ctx.L.skh_sketch_set_names(ss.h, arr))
    ss.names = names
    return ss


def fastx_to_sketches(ctx, files, params):
    """file_io.rs:141-252: one sketch per file, contigs < 500 bp skipped, files with no kept contig dropped,
    result sorted by file name (file_io.rs:250)."""
    files = sorted(files)
    genomes, names = [], []
    for f in files:
        recs = [(n, s) for n, s in read_fasta(f) if len(s) >= MIN_LENGTH_CONTIG]
        if recs:
            genomes.append(recs); names.append(f)
    return ctx.sketch_records(genomes, params, names)
