/*
 * skani_oracle.h -- C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a from-scratch CPU restatement of the reference skani v0.3.0 hot path
 * (sketch -> screen -> chain -> ANI/AF -> learned-ANI) used ONLY as the checker in tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under skani_amd/ may
 * include, link or call it.  Every function cites the reference file:line it follows
 * (paths relative to the reference checkout, bluenote-1577/skani @ v0.3.0).
 *
 * Parity pins (tests/test_oracle_golden.py): the 759 plasmid seed records + 81 markers inside
 * the reference's test_files/e.coli-o157.fasta.sketch, the three `search --median` triples of
 * test_results_versions/0.3.0:130-135, and the range assertions of tests/tests.rs.
 * UNPINNED by any reference test with available inputs: learned-ANI output values, bootstrap
 * CI bounds, argmax tie order inside a chain component, 'n' handling, --robust.
 */
#ifndef SKANI_ORACLE_H
#define SKANI_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_sketch ora_sketch;
typedef struct ora_model ora_model;

/* AniEstResult without strings (types.rs:559-582) */
typedef struct {
    float ani, af_query, af_ref, ci_lower, ci_upper, std;
    float q90_q, q90_r, q50_q, q50_r, q10_q, q10_r;
    uint32_t num_contigs_q, num_contigs_r, avg_chain_int_len, total_bases_covered;
} ora_ani_result;

/* CommandParams/MapParams subset that changes hot-path results (params.rs:74-123, chain.rs:88-142) */
typedef struct {
    double min_af;       /* <0 => 0.15 (chain.rs:100-107) */
    double both_min_af;  /* <=0 disabled */
    int robust, median;
} ora_map_opts;

/* optional per-stage dump of chain_seeds for debugging/parity of intermediate stages */
typedef struct {
    int switched;
    uint64_t n_anchors, n_chunks, n_intervals, n_accepted, n_estimates, n_qpos;
    uint64_t anchor_checksum;   /* order-sensitive FNV over sorted anchors */
    uint64_t interval_checksum; /* FNV over accepted intervals in acceptance order */
} ora_chain_stats;

ora_sketch* ora_sketch_new(uint32_t c, uint32_t k, uint32_t marker_c, const char* file_name);
void ora_sketch_free(ora_sketch*);
/* file_io.rs:176-230: contigs shorter than min_len (500) are skipped entirely; returns 1 if kept.
 * mode 0 = seeding.rs:225-323 (scalar), 1 = avx2_seeding.rs:33-272 semantics. */
int ora_sketch_add_contig(ora_sketch*, const uint8_t* seq, uint64_t len, int mode, uint64_t min_len);
/* file_io.rs:141-252 for a batch of files, sketched by `threads` workers (file_io.rs:147 par_iter): genome g = contigs
 * [genome_contig_off[g], genome_contig_off[g+1]) of seq[] / len[]; out[g] receives a new sketch named names[g].
 * mode 2 = the plain-C++ statement of mode 1 (mode 1 itself runs AVX2 intrinsics like the reference where the build has them). */
void ora_sketch_batch(uint32_t n_genomes, const uint64_t* genome_contig_off, const uint8_t* const* seq, const uint64_t* len,
                      uint32_t c, uint32_t k, uint32_t marker_c, const char* const* names, int mode, uint64_t min_len, int threads,
                      ora_sketch** out);
/* the same from plain FASTA files on disk, every file read by the worker that sketches it; out[i] = NULL for unreadable files and files without a kept contig */
void ora_sketch_files(uint32_t n_files, const char* const* paths, uint32_t c, uint32_t k, uint32_t marker_c, int mode, uint64_t min_len, int threads,
                      ora_sketch** out);
/* build a sketch from explicit arrays (golden fixture) */
ora_sketch* ora_sketch_from_arrays(uint32_t c, uint32_t k, uint32_t marker_c, const char* file_name,
                                   const uint32_t* seed, const uint32_t* pos, const uint32_t* ctgcanon,
                                   uint64_t n_pos, const uint64_t* markers, uint64_t n_markers,
                                   const uint32_t* contig_lengths, uint32_t n_contigs, uint64_t total_len);
uint64_t ora_sketch_n_positions(const ora_sketch*);
uint64_t ora_sketch_n_distinct(const ora_sketch*);
uint64_t ora_sketch_n_markers(const ora_sketch*);
uint32_t ora_sketch_n_contigs(const ora_sketch*);
uint64_t ora_sketch_total_len(const ora_sketch*);
/* sorted by (seed, contig, pos); arrays sized n_positions */
void ora_sketch_export_seeds(const ora_sketch*, uint32_t* seed, uint32_t* pos, uint32_t* ctgcanon);
/* position order (contig, pos) */
void ora_sketch_export_seeds_pos_order(const ora_sketch*, uint32_t* seed, uint32_t* pos, uint32_t* ctgcanon);
void ora_sketch_export_markers(const ora_sketch*, uint64_t* markers_sorted);
void ora_sketch_export_contig_lengths(const ora_sketch*, uint32_t* lens);

/* regression.rs:12-28: model = flat table produced by tools/extract_gbdt_model.py */
ora_model* ora_model_load(const char* path);
void ora_model_free(ora_model*);
float ora_model_predict(const ora_model*, const float feat[5]);

/* chain.rs:144-171 (+ regression.rs:30-64 when model != NULL) */
void ora_chain_seeds(const ora_sketch* ref, const ora_sketch* query, const ora_map_opts*,
                     const ora_model* model, ora_ani_result* out, ora_chain_stats* stats /*nullable*/);

/* screen.rs:84-142 */
int ora_check_markers_quickly(const ora_sketch* ref, const ora_sketch* query, double screen_val, int rescue_small);
/* screen.rs:148-189 (rule 0) / :39-77 (rule 2, no rescue) via the inverted index of :190-210.
 * Returns, for query sketch q against refs[0..n), the passing ref ids (unsorted -> sorted here). */
uint64_t ora_screen_refs(const ora_sketch* const* refs, uint32_t n_refs, const ora_sketch* query,
                         double identity, int rule, int rescue_small, uint32_t* out_ids /*cap n_refs*/);

/* triangle.rs:55-105: screen every row, chain pairs j>i that pass, keep ani>0.1.
 * out arrays sized by n*(n-1)/2 worst case are allocated by the caller (cap); returns number kept;
 * *n_chained = number of chain_seeds calls.  threads<=0 => hardware concurrency. */
uint64_t ora_triangle(const ora_sketch* const* sk, uint32_t n, double screen_val, int rescue_small,
                      const ora_map_opts*, const ora_model*, int threads,
                      uint32_t* out_i, uint32_t* out_j, ora_ani_result* out_res, uint64_t cap,
                      uint64_t* n_chained, uint64_t* n_screen_pass);
/* wall-clock seconds of the last ora_triangle call: marker index build, screen of all rows, chaining of the passing pairs */
void ora_triangle_phases(double* index_s, double* screen_s, double* chain_s);

/* search.rs:97-200 with every reference sketch resident (--keep-refs): for every query, refs that pass the marker screen -- use_index != 0:
 * screen_refs_indices through the inverted index built once over all refs (search.rs:60-66, parse.rs:960: more than 50 query files); else
 * check_markers_quickly(query, ref, screen_val, false) ref by ref (search.rs:122-131) -- are chained, chain_seeds(ref, query), and the results with
 * ani > 0.5 kept (search.rs:178).  Threads pull queries (the reference nests par_iters over queries, refs and hits).  Output in (query, ref) order;
 * returns the number kept; *n_chained = chain_seeds calls.  The phases' wall clock through ora_triangle_phases (index, screen, chain). */
uint64_t ora_search(const ora_sketch* const* refs, uint32_t n_refs, const ora_sketch* const* queries, uint32_t n_queries, double screen_val, int use_index,
                    const ora_map_opts*, const ora_model*, int threads, uint32_t* out_q, uint32_t* out_r, ora_ani_result* out_res, uint64_t cap,
                    uint64_t* n_chained);

/* pure helpers exposed for known-answer tests */
uint64_t ora_mm_hash64(uint64_t key);               /* types.rs:86-96 */
double ora_powi(double x, int n);                   /* compiler-rt __powidf2 as used by f64::powi */

#ifdef __cplusplus
}
#endif
#endif
