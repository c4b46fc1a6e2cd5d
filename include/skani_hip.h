/*
 * skani_hip.h -- C ABI of the MI355X-native skani hot path (libskani_hip.so).
 *
 * The reference (bluenote-1577/skani v0.3.0) has no FFI; its de-facto operator API is the pub Rust
 * library surface used by src/triangle.rs, src/dist.rs, src/search.rs and by pyskani.  Each entry point
 * below replaces one of those calls at BATCH granularity (per-pair FFI calls would serialise the GPU).
 * Citations are reference file:line.  The Rust-side binding a skani maintainer would add is shown in
 * INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; inputs are caller-owned and read-only for the call; outputs
 * allocated by the library are released with skh_free()/..._destroy(); every function returns 0 on success
 * or a negative skh_status and never unwinds across the boundary (skh_last_error() gives the message).
 * A context is bound to one GPU and must be driven by one host thread at a time.  Sketch sets are immutable
 * after creation and device resident.  There is NO CPU path: skh_ctx_create fails when no gfx950 device
 * is visible.
 */
#ifndef SKANI_HIP_H
#define SKANI_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SKH_OK = 0,
    SKH_ERR_INVALID = -1,   /* bad argument (k > 16, c > marker_c, ...: seeding.rs:239, params.rs:183-185) */
    SKH_ERR_DEVICE = -2,    /* HIP runtime failure / no device */
    SKH_ERR_NOMEM = -3,
    SKH_ERR_INTERNAL = -4,
    SKH_ERR_PEER = -5       /* skh_triangle_distributed: ANOTHER rank failed (or a collective was cut short because one did); that rank's own error says why */
} skh_status;

typedef struct skh_ctx skh_ctx;
typedef struct skh_genome_set skh_genome_set;   /* 2-bit packed contigs resident in HBM */
typedef struct skh_sketch_set skh_sketch_set;   /* Vec<Sketch> (types.rs:252-277) as device SoA + tables */

/* SketchParams (params.rs:136-196) + which seeding semantics to reproduce:
 * SKH_SEED_SCALAR = seeding.rs:225-323 (what non-x86 hosts run), SKH_SEED_AVX2 = avx2_seeding.rs:33-272
 * (what every x86-64+AVX2 host runs, file_io.rs:194-206) -- they differ in tail windows and N handling. */
enum { SKH_SEED_SCALAR = 0, SKH_SEED_AVX2 = 1 };
typedef struct {
    uint32_t c;            /* -c, default 125 */
    uint32_t k;            /* -k, default 15, <= 16 */
    uint32_t marker_c;     /* -m, default 1000, >= c */
    uint32_t seeding_mode; /* SKH_SEED_* */
} skh_sketch_params;

/* The CommandParams / MapParams fields that change chain_seeds results (chain.rs:88-142, params.rs:74-123).
 * Everything else in MapParams is a constant derived from the ref sketch's c,k. */
typedef struct {
    double min_af;        /* --min-af /100; < 0 selects the 0.15 default (chain.rs:100-107) */
    double both_min_af;   /* --both-min-af /100; <= 0 disabled (chain.rs:500-504) */
    uint8_t robust;       /* --robust */
    uint8_t median;       /* --median */
    uint8_t learned_ani;  /* apply regression.rs:30-64 (caller decides via regression.rs:8-10) */
    uint8_t compute_ci;   /* also fill ci_lower/ci_upper with the percentile bootstrap of chain.rs:57-86 */
} skh_map_params;

/* AniEstResult minus the strings (types.rs:559-582).  ani = NaN: no anchors/estimates (chain.rs:416-420);
 * ani = -1: aligned-fraction cut-off (chain.rs:500-517). */
typedef struct {
    float ani, af_query, af_ref, ci_lower, ci_upper, std;
    float q90_q, q90_r, q50_q, q50_r, q10_q, q10_r;
    uint32_t num_contigs_q, num_contigs_r, avg_chain_int_len, total_bases_covered;
} skh_ani_result;

/* per-pair stage sizes, for parity tests of intermediate stages and for profiling */
typedef struct {
    uint32_t switched, n_chunks, n_intervals, n_accepted, n_estimates, reserved;
    uint64_t n_anchors, n_qpos, anchor_checksum;
} skh_chain_stats;

/* ------------------------------------------------------------------ context */
int skh_ctx_create(int device, skh_ctx** out);
void skh_ctx_destroy(skh_ctx*);
const char* skh_last_error(const skh_ctx*);
void skh_free(void* p);
/* learned-ANI regression tables (regression.rs:12-28 picks by |c-125| < |c-200|); files are the flat tables
 * under skani_amd/data/ produced by tools/extract_gbdt_model.py */
int skh_load_models(skh_ctx*, const char* path_c125, const char* path_c200);

/* ------------------------------------------------------------------ ingest (file_io.rs:158-183) */
/* bases: ASCII contig bytes, contig i = bases[contig_off[i] .. contig_off[i+1]); contig_genome[i] = genome id
 * (non-decreasing).  The caller has already applied the >= 500 bp contig filter (file_io.rs:176) -- contigs
 * are indexed in the order given.  bases_on_device != 0: `bases` is a device pointer. */
int skh_genomes_pack(skh_ctx*, const uint8_t* bases, const uint64_t* contig_off, const uint32_t* contig_genome,
                     uint32_t n_contigs, uint32_t n_genomes, int bases_on_device, int seeding_mode,
                     skh_genome_set** out);
/* The same in batches, so that parsing, PCIe and packing overlap (file_io.rs:147 reads its files in parallel; here later files are parsed while
 * earlier ones are copied and packed).  skh_genomes_begin sizes the set: n_genomes genomes and at most max_bases bases, counting every contig's
 * length rounded up to a multiple of 64 (a parser that only knows its files' sizes announces 1.13 x their sum: contigs have at least 500 bases);
 * max_contigs is an estimate, the contig table grows when it is exceeded.
 * skh_genomes_append adds whole genomes: contig i of the batch = bases[contig_start[i] .. contig_start[i] + contig_len[i]) (gaps between contigs
 * are fine: parser threads may write their files into stretches of one buffer), contig_genome[i] = the genome's number in the set -- its
 * position in the caller's file order, whatever order the batches arrive in; a genome's contigs come together, in order, in ONE batch; a genome
 * that never arrives has no contigs.  The call returns when the batch is queued; with `bases` in pinned memory (skh_host_alloc) the copy runs
 * behind the caller's back and the buffer may be rewritten once skh_genomes_wait(ticket) has returned.  One thread at a time per context, like
 * every call -- except skh_genomes_wait, which any thread may call while another one appends or finishes (a wait holds its own reference to the batch's
 * event); every wait must have returned before skh_genomes_destroy.  A refused skh_genomes_append (a bad genome number, a genome in two batches, more
 * bases than announced) leaves the set as it was.  skh_genomes_finish completes the set (it then behaves like one from skh_genomes_pack).
 * Every skh_* call binds the calling thread to its context's device for its duration and puts the thread's previous current device back when it returns. */
void* skh_host_alloc(uint64_t bytes);   /* pinned host memory; NULL on failure */
void skh_host_free(void*);
int skh_genomes_begin(skh_ctx*, uint64_t max_bases, uint32_t max_contigs, uint32_t n_genomes, int seeding_mode, skh_genome_set** out);
int skh_genomes_append(skh_genome_set*, const uint8_t* bases, const uint64_t* contig_start, const uint64_t* contig_len, const uint32_t* contig_genome,
                       uint32_t n_contigs, int bases_on_device, uint64_t* ticket);
int skh_genomes_wait(skh_genome_set*, uint64_t ticket);
int skh_genomes_finish(skh_genome_set*);
void skh_genomes_destroy(skh_genome_set*);
uint64_t skh_genomes_total_bases(const skh_genome_set*);

/* ------------------------------------------------------------------ sketch: fmh_seeds / avx2_fmh_seeds
 * (seeding.rs:225, avx2_seeding.rs:33) for every contig + Sketch construction (types.rs:281-304).
 * genome_rank: optional lexicographic rank of each genome's file name, used only for the switch_qr tie
 * (chain.rs:20-22); NULL = genome index. */
int skh_sketch_genomes(skh_ctx*, const skh_genome_set*, const skh_sketch_params*, const uint32_t* genome_rank,
                       skh_sketch_set** out);
/* Same with flags.  SKH_SKETCH_DEFER_TABLES: seeding + marker sets only; the per-genome seed tables are built on first use (skh_chain_pairs*, skh_triangle*,
 * skh_sketch_sizes' n_distinct) or by skh_sketch_build_tables.  A rank of a distributed triangle sketches with this flag: it then indexes only the
 * sketches it ends up chaining (its own that stay + the ones it receives), not every genome it happened to read.  skh_triangle on such a set builds
 * the tables on one stream while the marker screen and the host's pair bookkeeping run beside them on the other (the set's screen index is made at
 * sketch time for that); SKH_SKETCH_NO_SCREEN_INDEX leaves the screen index out as well (a rank of a distributed triangle never screens its own set).
 * SKH_SKETCH_COMPACT: a set that is made to stay resident (the shards of a search database, search.rs:97-282): its seed tables take 1.5 instead of 2 home
 * slots per seed position and its list storage is cut to what the lists take -- about 21 instead of 31 bytes of HBM per seed position; probing such a
 * table walks slightly longer clusters.  Results are the same. */
enum { SKH_SKETCH_DEFER_TABLES = 1, SKH_SKETCH_NO_SCREEN_INDEX = 2, SKH_SKETCH_COMPACT = 4 };
int skh_sketch_genomes_ex(skh_ctx*, const skh_genome_set*, const skh_sketch_params*, const uint32_t* genome_rank, uint32_t flags,
                          skh_sketch_set** out);
int skh_sketch_build_tables(skh_ctx*, skh_sketch_set*);   /* no-op when they exist */
/* convenience = skh_genomes_pack + skh_sketch_genomes (the fastx_to_sketches body, file_io.rs:141-252) */
int skh_sketch_batch(skh_ctx*, const uint8_t* bases, const uint64_t* contig_off, const uint32_t* contig_genome,
                     uint32_t n_contigs, uint32_t n_genomes, const skh_sketch_params*, const uint32_t* genome_rank,
                     skh_sketch_set** out);
void skh_sketch_set_destroy(skh_sketch_set*);
/* Optional: the genomes' file names (copied).  Only consumer: the switch_qr tie `query_file_name > ref_file_name`
 * (chain.rs:20-22).  When both sets of a pair carry names the strings are compared; otherwise genome_rank is used, which
 * is only meaningful between sets ranked against a common ordering. */
int skh_sketch_set_names(skh_sketch_set*, const char* const* names);

/* sizes */
uint32_t skh_sketch_n_genomes(const skh_sketch_set*);
/* 1 when the set holds a genome of 2^31 - 8192 or more padded bases (total length + 8192 per contig: a human genome, or an assembly of 250,000
 * short contigs).  Such a set is "wide": it keeps the coordinates of its positions in 64 bits as well, and the 32-bit position records of its
 * wide genomes hold position indices.  Every entry point takes it like any other set and returns the same results; a chaining call runs the pairs
 * that involve a wide genome on 64-bit coordinates and the others as always (chain.hip).  What remains are the reference's own u32 fields
 * (types.rs:131-138): contigs below 2^32 bases, genomes below 2^30 seed positions.  skh_triangle_distributed: when any rank holds a wide set, the
 * ranks exchange positions in this header's (position in contig, contig << 1 | canonical) form -- 8 bytes per position instead of 4 -- and every
 * rank makes the set it chains through the import path. */
int skh_sketch_is_wide(const skh_sketch_set*);
int skh_sketch_sizes(const skh_sketch_set*, uint32_t g, uint64_t* n_pos, uint64_t* n_distinct, uint64_t* n_markers,
                     uint32_t* n_contigs, uint64_t* total_len);
/* flat export for serialisation / exchange between GPUs / parity checks.  Arrays are caller-allocated from
 * skh_sketch_sizes; seeds come in position order (contig, pos); ctgcanon = contig<<1 | canonical
 * (types.rs:124-143); markers sorted ascending (marker_seeds is a set, types.rs:272). */
int skh_sketch_export(const skh_sketch_set*, uint32_t g, uint32_t* seed, uint32_t* pos, uint32_t* ctgcanon,
                      uint64_t* markers, uint32_t* contig_lengths);
/* build a set from exported arrays of n_genomes genomes (concatenated; *_off have n_genomes+1 entries).
 * This is sketches_from_sketch (file_io.rs:680-729) for an in-memory format. */
int skh_sketch_import(skh_ctx*, const skh_sketch_params*, uint32_t n_genomes, const uint64_t* pos_off,
                      const uint32_t* seed, const uint32_t* pos, const uint32_t* ctgcanon, const uint64_t* marker_off,
                      const uint64_t* markers, const uint64_t* contig_off, const uint32_t* contig_lengths,
                      const uint64_t* total_len, const uint32_t* genome_rank, skh_sketch_set** out);

/* Whole-set ("flat") export / import: the exchange format between GPUs.  Totals first, then one call moves everything:
 * the four big arrays (seed, pos, ctgcanon: position order, all genomes concatenated; markers: sorted per genome) go to /
 * come from DEVICE memory when arrays_on_device != 0 (e.g. torch tensors handed to RCCL), host memory otherwise; the small
 * per-genome tables (offsets with n_genomes+1 entries, contig lengths, total lengths, ranks) are always host arrays.
 * Any output pointer may be NULL to skip that array. */
int skh_sketch_totals(const skh_sketch_set*, uint64_t* n_pos, uint64_t* n_markers, uint64_t* n_contigs);
int skh_sketch_export_flat(const skh_sketch_set*, int arrays_on_device, uint32_t* seed, uint32_t* pos, uint32_t* ctgcanon, uint64_t* markers,
                           uint64_t* pos_off, uint64_t* marker_off, uint64_t* contig_off, uint32_t* contig_lengths, uint64_t* total_len,
                           uint32_t* genome_rank);
int skh_sketch_import_flat(skh_ctx*, const skh_sketch_params*, uint32_t n_genomes, int arrays_on_device, const uint64_t* pos_off,
                           const uint32_t* seed, const uint32_t* pos, const uint32_t* ctgcanon, const uint64_t* marker_off,
                           const uint64_t* markers, const uint64_t* contig_off, const uint32_t* contig_lengths, const uint64_t* total_len,
                           const uint32_t* genome_rank, skh_sketch_set** out);

/* ------------------------------------------------------------------ screen (screen.rs) */
enum { SKH_SCREEN_REFS = 0,            /* screen_refs, screen.rs:148-189 (triangle, dist with index) */
       SKH_SCREEN_QUICK = 1,           /* check_markers_quickly, screen.rs:84-142 */
       SKH_SCREEN_REFS_INDICES = 2 };  /* screen_refs_indices, screen.rs:39-77 (search with index) */
/* queries == NULL: triangle mode -- pairs (i, j>i) of `refs` with row genome i as the "query" of the rule
 * (triangle.rs:71-90).  Otherwise all (ref r, query q) pairs passing the rule.  identity == 0 selects the
 * 0.80 default (triangle.rs:34-42).  Output pair arrays (library-allocated, skh_free) are sorted by
 * (first, second): first = row/query index, second = ref index (triangle: i, j). */
int skh_screen(skh_ctx*, const skh_sketch_set* refs, const skh_sketch_set* queries, double identity, int rule,
               int rescue_small, uint32_t** pair_first, uint32_t** pair_second, uint64_t* n_pairs);

/* ------------------------------------------------------------------ chain: chain_seeds (chain.rs:144-171)
 * for each pair p: out[p] = chain_seeds(refs[pair_ref[p]], queries[pair_query[p]], map_params_from_sketch(ref)).
 * queries == NULL means queries = refs.  stats may be NULL. */
int skh_chain_pairs(skh_ctx*, const skh_sketch_set* refs, const skh_sketch_set* queries, const uint32_t* pair_ref,
                    const uint32_t* pair_query, uint64_t n_pairs, const skh_map_params*, skh_ani_result* out,
                    skh_chain_stats* stats);

/* Same, for a database kept as several resident sketch sets (shards): pair p = (ref_sets[pair_set[p]][pair_ref[p]],
 * queries[pair_query[p]]).  One call chains hits from every shard (search.rs:150-180 with all references resident). */
int skh_chain_pairs_multi(skh_ctx*, const skh_sketch_set* const* ref_sets, uint32_t n_ref_sets, const skh_sketch_set* queries,
                          const uint32_t* pair_set, const uint32_t* pair_ref, const uint32_t* pair_query, uint64_t n_pairs,
                          const skh_map_params*, skh_ani_result* out);

/* The triangle's screen (triangle.rs:55-90) for rows i in [row0, row0 + n_rows) only: pairs (i, j), j > i, of one set that
 * pass screen_refs with sketch i as the query.  This is one GPU's share when the rows of a large collection are
 * block-distributed over several GPUs that all hold the marker sets. */
int skh_screen_rows(skh_ctx*, const skh_sketch_set* set, uint32_t row0, uint32_t n_rows, double identity, int rescue_small,
                    uint32_t** pair_i, uint32_t** pair_j, uint64_t* n_pairs);

/* The triangle's screen cut by KEY RANGE instead of by rows -- how skh_triangle_distributed shares it among GPUs that all hold the marker sets, exposed for
 * hosts that run their own distribution.  skh_screen_part: the non-zero cells (i < j, count of shared markers; one 64-bit word each) of the triangle's count matrix over the
 * markers whose leading 16 bases fall into part `part` of `n_parts` -- a marker's incidences all lie in one part, so the parts' cells add up to the full
 * matrix; skh_screen_from_cells: the candidate pairs (screen_refs with sketch i as the query, triangle.rs:71-90) from the concatenated cells of all parts.
 * Both need n_genomes^2 counters of scratch (an error beyond the screen's budget); arrays are library-allocated (skh_free). */
int skh_screen_part(skh_ctx*, const skh_sketch_set* set, uint32_t part, uint32_t n_parts, uint64_t** cells, uint64_t* n_cells);   /* a cell: i << 43 | j << 22 | count */
int skh_screen_from_cells(skh_ctx*, const skh_sketch_set* set, const uint64_t* cells, uint64_t n_cells, double identity, int rescue_small,
                          uint32_t** pair_i, uint32_t** pair_j, uint64_t* n_pairs);

/* ------------------------------------------------------------------ triangle body (triangle.rs:55-105):
 * screen rows, chain pairs j>i, keep ani > 0.1.  part/n_parts shard the screened pair list round-robin
 * across GPUs (part r takes pairs r, r+n_parts, ...); results come back sorted by (i,j). */
int skh_triangle(skh_ctx*, const skh_sketch_set*, double identity, int rescue_small, const skh_map_params*,
                 uint32_t part, uint32_t n_parts, uint32_t** out_i, uint32_t** out_j, skh_ani_result** out_res,
                 uint64_t* n_kept, uint64_t* n_chained);

/* ------------------------------------------------------------------ the triangle over several GPUs (one process per GPU)
 * The reference parallelises triangle.rs:71-105 with a work-stealing thread pool over one shared Vec<Sketch>.  Here every GPU (rank) sketches
 * its own block of the genomes and skh_triangle_distributed() does the rest below this boundary:
 *   1. one all-gather of per-rank tables (who holds which genomes, their sizes, contig lengths, and per genome how many of its markers fall into each rank's part
 *      of the key range), then ONE all-to-all of marker sets by key range: every rank receives its own part of every genome's sorted set (device memory; 1/W of the
 *      bytes an all-gather of the sets would move).  Only the row form of step 2 all-gathers the whole sets;
 *   2. the screen is cut by KEY RANGE: rank r sorts and walks the (marker, genome) incidences whose leading 16 bases fall into its part of the key range and
 *      gets partial counts for every cell; the non-zero cells are gathered on the device and every rank adds them up and applies screen_refs' rule to
 *      every row itself -- the same candidate list on every rank, nothing to gather (a collection whose N x N count matrix is beyond the screen's budget is
 *      cut by rows instead and the candidate lists are gathered);
 *   3. every rank computes the SAME assignment of pairs to ranks: connected components of the candidate graph (clusters of related genomes) are kept
 *      whole, components too large for an even split are cut into (row-block x column-block) tiles, and the units are dealt out by estimated cost,
 *      longest first -- the balanced stand-in for the reference's work stealing; it does not depend on the order of the genomes;
 *   4. every rank receives, point-to-point and asynchronously, exactly the sketches its units need and it does not own (seed + position arrays, device
 *      memory), chains the pairs of its own sketches meanwhile, builds the seed tables of what arrived and chains the rest;
 *   5. the result rows are gathered: on every rank (skh_triangle_distributed) or on rank 0 only (SKH_DIST_ROWS_TO_ROOT).
 * Global genome index = (number of genomes on lower ranks) + local index; ranks may hold different numbers of genomes (also none).
 * genome_rank of the local set must be the genome's rank in ONE ordering common to all ranks, consistent with the file names' order where
 * names were set: inside this call the switch_qr tie (chain.rs:20-22) always goes by genome_rank -- names are not exchanged, and which rank
 * chains a pair must not decide its orientation.  A failure on one rank (out of memory, a table overflow, ...) is agreed on at the next
 * exchange point: EVERY rank returns an error, none is left waiting in a collective.
 * A communicator either runs on RCCL (device buffers over xGMI; the library loads librccl.so.1 itself) or on host-memory collectives the
 * caller supplies (MPI, gloo, ...: the library stages device data through host buffers) -- the latter is what the CPU tests use. */
typedef struct skh_comm skh_comm;
#define SKH_COMM_ID_BYTES 128
/* rank 0 creates the id and hands it to the other ranks by its own means (the host's launcher / MPI / a file) */
int skh_comm_unique_id(uint8_t id[SKH_COMM_ID_BYTES]);
int skh_comm_create_rccl(skh_ctx*, const uint8_t id[SKH_COMM_ID_BYTES], int rank, int world, skh_comm** out);
/* host-memory collectives: all_gather -- every rank contributes `bytes` bytes, recv gets world * bytes in rank order;
 * all_to_all_v -- send_cnt[r] bytes at send + send_off[r] go to rank r, recv_cnt[r] bytes from rank r land at recv + recv_off[r].
 * Both return 0 on success. */
typedef struct {
    void* user;
    int (*all_gather)(void* user, const void* send, void* recv, uint64_t bytes);
    int (*all_to_all_v)(void* user, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                        const uint64_t* recv_off);
} skh_host_collectives;
int skh_comm_create_host(skh_ctx*, const skh_host_collectives*, int rank, int world, skh_comm** out);
void skh_comm_destroy(skh_comm*);
/* Collective: a small all-gather of host memory, one of device memory and a sketch-exchange-shaped all-to-all (blocking and asynchronous) through the
 * communicator, every byte checked.  Meant to be called once after creation, where all ranks can still agree on another transport. */
int skh_comm_selftest(skh_ctx*, skh_comm*);
/* what this rank did in the last distributed triangle (balance / traffic, for tests and bench.py) */
typedef struct {
    uint64_t n_genomes_total, n_candidate_pairs_total;   /* the whole collection */
    uint64_t n_pairs_mine, n_units_mine, n_units_total;  /* this rank's share of the chaining */
    uint64_t cost_mine, cost_total;                      /* estimated chaining cost (sum of both genomes' marker counts per pair) */
    uint64_t n_genomes_received, bytes_received, bytes_sent;   /* sketches that crossed ranks */
    uint64_t screen_row_begin, screen_row_end;           /* this rank's rows of the screen when it is cut by rows (screen_by_key_range == 0) */
    uint64_t n_pairs_home;                               /* pairs of this rank with both sketches its own: chained while the other sketches travel */
    uint64_t exchange_async_us, exchange_wait_us;        /* the sketch exchange from start to last byte, and the part of it the chaining had to wait for */
    uint64_t screen_by_key_range;                        /* 1: every rank screened a W-th of the markers' key range for all cells (collections whose count matrix fits); 0: its rows */
    uint64_t marker_bytes_received;                      /* marker sets that reached this rank from the others: its own part of the key range (key-range form) or everything (row form) */
} skh_dist_stats;
/* Collective: every rank of the communicator calls it with its own local set.  Results (global indices, sorted by (i, j), ani > 0.1 as
 * triangle.rs:99) are returned on EVERY rank; n_chained = candidate pairs chained over all ranks.  stats may be NULL. */
int skh_triangle_distributed(skh_ctx*, skh_comm*, const skh_sketch_set* local, double identity, int rescue_small, const skh_map_params*,
                             uint32_t** out_i, uint32_t** out_j, skh_ani_result** out_res, uint64_t* n_kept, uint64_t* n_chained,
                             skh_dist_stats* stats);
/* The same with flags.  SKH_DIST_ROWS_TO_ROOT: the result rows travel to rank 0 only (triangle.rs:99-105 collects them in one place, and a host whose rank 0
 * writes the matrix needs them nowhere else): rank 0 returns the whole triangle, every other rank the rows of the pairs it chained itself (both in (i, j)
 * order).  n_chained is the total on every rank either way. */
enum { SKH_DIST_ROWS_TO_ROOT = 1 };
int skh_triangle_distributed_ex(skh_ctx*, skh_comm*, const skh_sketch_set* local, double identity, int rescue_small, const skh_map_params*, uint32_t flags,
                                uint32_t** out_i, uint32_t** out_j, skh_ani_result** out_res, uint64_t* n_kept, uint64_t* n_chained,
                                skh_dist_stats* stats);

/* The assignment step of skh_triangle_distributed on its own (host only, no communicator): owner[p] = rank that would chain candidate pair
 * (pair_i[p], pair_j[p]); weight[g] = cost proxy of genome g (the distributed triangle uses the marker count); holder[g] = rank whose GPU
 * holds genome g's sketch (NULL: no preference).  Lets a host inspect balance and traffic before committing GPUs to a collection. */
int skh_plan_pairs(uint32_t n_genomes, const uint32_t* pair_i, const uint32_t* pair_j, uint64_t n_pairs, const uint64_t* weight, const uint32_t* holder,
                   int world, uint8_t* owner);

/* Device memory the library holds (all contexts of the process): live_bytes = in use by sketch sets, genome sets, models and scratch; idle_bytes = freed
 * blocks kept for reuse by its caching allocator (up to SKH_TUNE_ALLOC_CACHE_BYTES; default: a third of the device, beyond 8 GiB only while an eighth of the
 * device is free -- they look "used" to the driver).  trim != 0 hands the
 * idle blocks back to the driver first.  Either pointer may be NULL. */
int skh_device_memory(uint64_t* live_bytes, uint64_t* idle_bytes, int trim);

/* last-call timing breakdown in milliseconds (HIP events on the library's stream), for bench.py */
typedef struct { float pack_ms, seed_ms, sketch_build_ms, screen_ms, chain_ms, seed_kernel_ms; uint32_t seed_kernel_launches; float exchange_ms; } skh_timings;
int skh_get_timings(const skh_ctx*, skh_timings* out);

#ifdef __cplusplus
}
#endif
#endif
