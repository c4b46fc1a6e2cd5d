"""ctypes prototypes for the C ABI declared in include/skani_hip.h."""
import ctypes as C

import numpy as np


class SketchParams(C.Structure):
    """skh_sketch_params (reference SketchParams, params.rs:136-196)."""
    _fields_ = [("c", C.c_uint32), ("k", C.c_uint32), ("marker_c", C.c_uint32), ("seeding_mode", C.c_uint32)]


class MapParams(C.Structure):
    """skh_map_params (the CommandParams fields consumed by map_params_from_sketch, chain.rs:88-142)."""
    _fields_ = [("min_af", C.c_double), ("both_min_af", C.c_double), ("robust", C.c_uint8), ("median", C.c_uint8),
                ("learned_ani", C.c_uint8), ("compute_ci", C.c_uint8)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("pack_ms", "seed_ms", "sketch_build_ms", "screen_ms", "chain_ms", "seed_kernel_ms")] + \
               [("seed_kernel_launches", C.c_uint32), ("exchange_ms", C.c_float)]


RESULT_DTYPE = np.dtype([(n, np.float32) for n in
                         ("ani", "af_query", "af_ref", "ci_lower", "ci_upper", "std",
                          "q90_q", "q90_r", "q50_q", "q50_r", "q10_q", "q10_r")] +
                        [(n, np.uint32) for n in ("num_contigs_q", "num_contigs_r", "avg_chain_int_len", "total_bases_covered")])

STATS_DTYPE = np.dtype([(n, np.uint32) for n in ("switched", "n_chunks", "n_intervals", "n_accepted", "n_estimates", "reserved")] +
                       [(n, np.uint64) for n in ("n_anchors", "n_qpos", "anchor_checksum")])

class DistStats(C.Structure):
    """skh_dist_stats"""
    _fields_ = [(n, C.c_uint64) for n in ("n_genomes_total", "n_candidate_pairs_total", "n_pairs_mine", "n_units_mine", "n_units_total", "cost_mine", "cost_total",
                                          "n_genomes_received", "bytes_received", "bytes_sent", "screen_row_begin", "screen_row_end",
                                          "n_pairs_home", "exchange_async_us", "exchange_wait_us", "screen_by_key_range", "marker_bytes_received")]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
ALL_TO_ALL_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))


class HostCollectives(C.Structure):
    """skh_host_collectives"""
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("all_to_all_v", ALL_TO_ALL_V_FN)]


EXPORTS = ["skh_ctx_create", "skh_ctx_destroy", "skh_last_error", "skh_free", "skh_load_models", "skh_genomes_pack",
           "skh_host_alloc", "skh_host_free", "skh_genomes_begin", "skh_genomes_append", "skh_genomes_wait", "skh_genomes_finish",
           "skh_genomes_destroy", "skh_genomes_total_bases", "skh_sketch_genomes", "skh_sketch_genomes_ex", "skh_sketch_build_tables", "skh_sketch_batch", "skh_sketch_set_destroy",
           "skh_sketch_set_names", "skh_sketch_n_genomes", "skh_sketch_is_wide", "skh_sketch_sizes", "skh_sketch_export", "skh_sketch_import", "skh_sketch_totals", "skh_sketch_export_flat", "skh_sketch_import_flat", "skh_screen", "skh_screen_rows", "skh_screen_part", "skh_screen_from_cells", "skh_chain_pairs", "skh_chain_pairs_multi",
           "skh_triangle", "skh_get_timings", "skh_device_memory", "skh_comm_unique_id", "skh_comm_create_rccl", "skh_comm_create_host", "skh_comm_destroy", "skh_comm_selftest", "skh_triangle_distributed", "skh_triangle_distributed_ex", "skh_plan_pairs"]
RCCL_ONLY = ("skh_comm_unique_id", "skh_comm_create_rccl")   # absent from the test-only simulator build (tests/emu)


def load(path):
    L = C.CDLL(path)
    vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    pp = C.POINTER(vp)
    L.skh_ctx_create.restype = i32; L.skh_ctx_create.argtypes = [i32, pp]
    L.skh_ctx_destroy.restype = None; L.skh_ctx_destroy.argtypes = [vp]
    L.skh_last_error.restype = C.c_char_p; L.skh_last_error.argtypes = [vp]
    L.skh_free.restype = None; L.skh_free.argtypes = [vp]
    L.skh_load_models.restype = i32; L.skh_load_models.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.skh_genomes_pack.restype = i32; L.skh_genomes_pack.argtypes = [vp, vp, vp, vp, u32, u32, i32, i32, pp]
    L.skh_host_alloc.restype = vp; L.skh_host_alloc.argtypes = [u64]
    L.skh_host_free.restype = None; L.skh_host_free.argtypes = [vp]
    L.skh_genomes_begin.restype = i32; L.skh_genomes_begin.argtypes = [vp, u64, u32, u32, i32, pp]
    L.skh_genomes_append.restype = i32; L.skh_genomes_append.argtypes = [vp, vp, vp, vp, vp, u32, i32, C.POINTER(u64)]
    L.skh_genomes_wait.restype = i32; L.skh_genomes_wait.argtypes = [vp, u64]
    L.skh_genomes_finish.restype = i32; L.skh_genomes_finish.argtypes = [vp]
    L.skh_genomes_destroy.restype = None; L.skh_genomes_destroy.argtypes = [vp]
    L.skh_genomes_total_bases.restype = u64; L.skh_genomes_total_bases.argtypes = [vp]
    L.skh_sketch_genomes.restype = i32; L.skh_sketch_genomes.argtypes = [vp, vp, C.POINTER(SketchParams), vp, pp]
    L.skh_sketch_genomes_ex.restype = i32; L.skh_sketch_genomes_ex.argtypes = [vp, vp, C.POINTER(SketchParams), vp, u32, pp]
    L.skh_sketch_build_tables.restype = i32; L.skh_sketch_build_tables.argtypes = [vp, vp]
    L.skh_sketch_batch.restype = i32; L.skh_sketch_batch.argtypes = [vp, vp, vp, vp, u32, u32, C.POINTER(SketchParams), vp, pp]
    L.skh_sketch_set_destroy.restype = None; L.skh_sketch_set_destroy.argtypes = [vp]
    L.skh_sketch_set_names.restype = i32; L.skh_sketch_set_names.argtypes = [vp, C.POINTER(C.c_char_p)]
    L.skh_sketch_n_genomes.restype = u32; L.skh_sketch_n_genomes.argtypes = [vp]
    L.skh_sketch_is_wide.restype = C.c_int; L.skh_sketch_is_wide.argtypes = [vp]
    L.skh_sketch_sizes.restype = i32
    L.skh_sketch_sizes.argtypes = [vp, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), C.POINTER(u64)]
    L.skh_sketch_export.restype = i32; L.skh_sketch_export.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    L.skh_sketch_import.restype = i32
    L.skh_sketch_import.argtypes = [vp, C.POINTER(SketchParams), u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, pp]
    L.skh_sketch_totals.restype = i32; L.skh_sketch_totals.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.skh_sketch_export_flat.restype = i32; L.skh_sketch_export_flat.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.skh_sketch_import_flat.restype = i32
    L.skh_sketch_import_flat.argtypes = [vp, C.POINTER(SketchParams), u32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, pp]
    L.skh_screen.restype = i32; L.skh_screen.argtypes = [vp, vp, vp, dbl, i32, i32, pp, pp, C.POINTER(u64)]
    L.skh_screen_rows.restype = i32; L.skh_screen_rows.argtypes = [vp, vp, u32, u32, dbl, i32, pp, pp, C.POINTER(u64)]
    L.skh_screen_part.restype = i32; L.skh_screen_part.argtypes = [vp, vp, u32, u32, pp, C.POINTER(u64)]
    L.skh_screen_from_cells.restype = i32; L.skh_screen_from_cells.argtypes = [vp, vp, vp, u64, dbl, i32, pp, pp, C.POINTER(u64)]
    L.skh_chain_pairs.restype = i32; L.skh_chain_pairs.argtypes = [vp, vp, vp, vp, vp, u64, C.POINTER(MapParams), vp, vp]
    L.skh_chain_pairs_multi.restype = i32; L.skh_chain_pairs_multi.argtypes = [vp, vp, u32, vp, vp, vp, vp, u64, C.POINTER(MapParams), vp]
    L.skh_triangle.restype = i32
    L.skh_triangle.argtypes = [vp, vp, dbl, i32, C.POINTER(MapParams), u32, u32, pp, pp, pp, C.POINTER(u64), C.POINTER(u64)]
    L.skh_get_timings.restype = i32; L.skh_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.skh_device_memory.restype = i32; L.skh_device_memory.argtypes = [C.POINTER(u64), C.POINTER(u64), i32]
    L.skh_comm_create_host.restype = i32; L.skh_comm_create_host.argtypes = [vp, C.POINTER(HostCollectives), i32, i32, pp]
    L.skh_comm_destroy.restype = None; L.skh_comm_destroy.argtypes = [vp]
    L.skh_comm_selftest.restype = i32; L.skh_comm_selftest.argtypes = [vp, vp]
    L.skh_triangle_distributed.restype = i32
    L.skh_triangle_distributed.argtypes = [vp, vp, vp, dbl, i32, C.POINTER(MapParams), pp, pp, pp, C.POINTER(u64), C.POINTER(u64), C.POINTER(DistStats)]
    L.skh_triangle_distributed_ex.restype = i32
    L.skh_triangle_distributed_ex.argtypes = [vp, vp, vp, dbl, i32, C.POINTER(MapParams), u32, pp, pp, pp, C.POINTER(u64), C.POINTER(u64), C.POINTER(DistStats)]
    L.skh_plan_pairs.restype = i32; L.skh_plan_pairs.argtypes = [u32, vp, vp, u64, vp, vp, i32, vp]
    if hasattr(L, "skh_comm_create_rccl"):
        L.skh_comm_unique_id.restype = i32; L.skh_comm_unique_id.argtypes = [vp]
        L.skh_comm_create_rccl.restype = i32; L.skh_comm_create_rccl.argtypes = [vp, vp, i32, i32, pp]
    return L
