// internal.h -- host-side declarations shared by the translation units of libskani_hip.so.
#pragma once
#include <chrono>
#include <functional>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "../../include/skani_hip.h"
#include "common.h"

namespace skh {

// Stream-ordered scratch: bump allocation out of large device chunks; everything handed out is recycled
// when the owning API call finishes (reset()).  Avoids hipMalloc/hipFree (implicit syncs) between kernels.
struct Arena {
    struct Chunk { char* p; size_t cap, off; };
    std::vector<Chunk> chunks;
    size_t min_chunk = (size_t)256 << 20;
    ~Arena() { for (auto& c : chunks) dfree(c.p); }
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        for (auto& c : chunks) if (c.cap - c.off >= bytes) { void* r = c.p + c.off; c.off += bytes; return r; }
        size_t cap = bytes > min_chunk ? bytes : min_chunk;
        Chunk c{(char*)dmalloc(cap), cap, bytes};
        chunks.push_back(c);
        return c.p;
    }
    template <class T> T* get(size_t n) { return (T*)take(n * sizeof(T)); }
    std::vector<size_t> mark() const { std::vector<size_t> m; for (auto& c : chunks) m.push_back(c.off); return m; }
    void rewind(const std::vector<size_t>& m) {   // callers sync the stream first
        for (size_t i = 0; i < chunks.size(); i++) chunks[i].off = i < m.size() ? m[i] : 0;
    }
    size_t capacity() const { size_t t = 0; for (auto& c : chunks) t += c.cap; return t; }
    void reset() {  // keep the largest chunk, free the rest (callers sync the stream first)
        if (chunks.size() > 1) {
            size_t tot = 0; for (auto& c : chunks) { tot += c.cap; dfree(c.p); }
            chunks.clear(); Chunk c{(char*)dmalloc(tot), tot, 0}; chunks.push_back(c);
        } else for (auto& c : chunks) c.off = 0;
    }
    void release_all() { for (auto& c : chunks) dfree(c.p); chunks.clear(); }
};

struct GbdtModel {  // flat table from tools/extract_gbdt_model.py (regression.rs:12-28)
    uint32_t n_trees = 0, n_feat = 0, n_nodes = 0; float shrinkage = 0, bias = 0;
    struct Node { int32_t feat; float thr, pred; int32_t left, right; };
    DBuf<uint32_t> off; DBuf<Node> nodes;
    bool loaded() const { return n_trees != 0; }
};

}  // namespace skh

// Scratch budgets that decide how work is split into launches/batches.  The defaults suit 288 GB of HBM; tests shrink them
// (SKH_TUNE_* environment variables, read at context creation) to drive the multi-batch paths with small inputs.
struct skh_tunables {
    uint64_t seed_scratch_bytes = (uint64_t)16 << 30;   // capped tile scratch per seeding launch (and at most a quarter of the device's free memory unless set by SKH_TUNE_SEED_SCRATCH_BYTES)
    bool seed_scratch_fixed = false;                    // set from the environment: taken as it is
    uint32_t seed_tile_cap = 0;                         // seeds a tile may list in the capped scratch (0 = 4x the expected number; tests use few: every tile then only counts and is re-run with full capacity)
    uint64_t screen_cells = (uint64_t)2 << 30;          // u32 counters of the screen's dense row block
    uint64_t chain_anchors = (uint64_t)512 << 20;       // anchors per chain batch (~55 B of scratch each: anchors, candidate intervals, 32 B per candidate slot of the pairs that may select in global memory)
    uint32_t chain_super_tiles = 1u << 20;              // join tiles per count pass (6 KiB of probe records each)
    uint32_t wide_sweep_dp = 0;                         // 1: a run on 64-bit coordinates chains with the wave-per-chunk sweep kernel whatever its band (the form before round 5; tests)
    uint32_t build_resalt_all = 0;                      // 1: the table build treats every genome as crowded once (tests: the second salt, and what was derived from the set ahead of the build's end)
    uint32_t screen_sort_radix = 0;                     // 1: the screen's incidence keys go through the device-wide radix sort (the form before round 5; tests, A/B runs)
    uint32_t skeys_avg = 1400;                          // keys per bucket the incidence sort aims at (screen_keys.hip)
    uint32_t skeys_cap = 0;                             // != 0: a lower limit than SKEYS_CAP_MAX on the keys of a bucket sorted in LDS (tests of the radix-sort way out)
    uint32_t screen_cells_dense = 0;                    // 1: the gathered cells of the key-range screen are added up in the dense N x N matrix (the form before round 5; tests)
    uint32_t screen_count_rows = 1;                      // triangle screen: the lanes of a marker's group walk it together, one row of the count matrix per instruction (0: lane e pairs with e + s, the form before round 6; A/B runs, tests)
    uint32_t screen_col_order = 1;                      // triangle screen with screen_count_rows: the count matrix's columns grouped by the clusters the incidences tie together (0: columns = genomes in collection order; 2: also for key-range parts too small for it to pay; A/B runs, tests)
    uint32_t marker_gate = 1;                           // the sketch call launches the marker sets' kernel first and lets build_tables_kernel wait for it (0: both start together, the form before round 6; A/B runs)
    uint32_t screen_planes = 8;                         // triangle screen: copies of the count matrix, one per XCD (1 = a single device-scope copy)
    uint32_t join_bitmap_words = 8192;                  // LDS words (32 KB) the join may spend on a probed sketch's bucket bitmap; larger bitmaps are not staged
    uint32_t build_slice_max = 0;                       // table slices per genome the slice-list kernel handles (0 = 8192; tests use 1: larger genomes' slices re-scan)
    uint32_t marker_lds_max = 0;                        // raw markers per genome the in-LDS marker-set kernel takes (0 = 8192; tests use few to force the device-wide path)
    uint32_t build_match_cap = 0;                       // positions a table slice may list in LDS on the first attempt (0 = as many as the slice has home slots; tests use few to force the re-scanning path)
    uint32_t greedy_len_limit = 0x10000;               // chain intervals at least this long on either axis send their pair to the general selection kernel (tests use a small value to drive that hand-over)
    uint32_t greedy_big_min = 2049;                    // candidate intervals from which a pair's selection runs in global memory (greedy_big_kernel); tests use small values
    uint64_t scan_one_max = ~0ull, scan_two_max = ~0ull; // tests: the largest arrays the one-launch / two-launch prefix sums take (scan.hip; smaller values push small inputs through the other forms)
    uint32_t dist_key_range_w1 = 0;                     // tests: a world of one screens by key range too (the cell gather on device buffers through the transport)
    uint32_t dist_fail = 0;                             // tests: the n-th local phase of a distributed triangle fails on this rank (0 = never)
    uint32_t chain_dp_lds_slots = 8;                    // live-chain slots per DP lane kept in LDS (8, or 1 to exercise the spill path)
    uint64_t wide_span = (1ull << 31) - 8192;           // a genome of at least this many padded bases makes its sketch set "wide" (tests use small values to run everything through the 64-bit path)
};

namespace skh {
struct PendingSort {
    DBuf<uint64_t> raw; DBuf<char> tmp; DevEvent ev; std::mutex mu;                  // raw: the keys in their buckets; tmp: bucket counters, offsets, cursors, range bounds (screen_keys.hip)
    DBuf<uint32_t> mat; DBuf<uint64_t> work; DBuf<char> sort_tmp;                    // the column order made behind the sort (screen.hip queue_column_order_ahead): its N x N sample matrix, union-find parents and labels
    void release() { std::lock_guard<std::mutex> lk(mu); raw.release(); tmp.release(); mat.release(); work.release(); sort_tmp.release(); }   // (only once the event is done)
};
}  // namespace skh

struct skh_ctx {
    skh_tunables tune;
    int device = 0;
    devStream_t stream{};
    devStream_t stream2{};                               // second queue: the marker sets are built there while the seed tables are built on `stream`
    std::string err;
    skh::Arena arena;
    skh::GbdtModel model_c125, model_c200;
    skh_timings timings{};
    skh::PinBuf pin_pairs;                               // the chaining's pair descriptors (host side)
    skh::PinBuf pin_results;                             // skh_triangle's result rows on their way back (pinned: the read-back is one DMA, and no fresh pages are touched per call)
    skh::PinRing ring;                                   // pinned staging of this context's small uploads (dev.h h2d); entry points bind it to their thread
    skh::DBuf<uint32_t> scan_ticket;                     // two counters (one per stream), zero between scans (scan.hip)
    bool screen_planes_checked = false;                  // the per-XCD count planes of the triangle screen passed their self-test (screen.hip)
    // the key-range screen's count matrix of a large collection (one plane, N x N words) stays with the context, all zero between calls: the kernel that emits the
    // non-zero cells puts them back to zero, so no call zeroes 4 N^2 bytes (screen.hip screen_partial_cells_dev); part_cnt_clean is false while a call is in between
    skh::DBuf<uint32_t> part_cnt; bool part_cnt_clean = false;
    std::vector<std::shared_ptr<skh::PendingSort>> pending_sorts;   // index sorts this context queued and did not wait for (screen.hip reap_pending_sorts)
};

namespace skh { struct Transport; }
struct skh_comm { skh::Transport* t = nullptr; ~skh_comm(); };   // destructor in dist.hip (Transport is defined below)

struct skh_genome_set {
    skh_ctx* ctx = nullptr;
    int seeding_mode = 0;
    uint32_t n_genomes = 0, n_contigs = 0;
    uint64_t n_words = 0, total_bases = 0;
    std::vector<skh::ContigDesc> contigs;          // host copy
    std::vector<uint64_t> genome_contig_off;       // n_genomes+1
    skh::DBuf<uint32_t> packed;                    // 2-bit MSB-first, 16 bases per word
    skh::DBuf<uint32_t> nmask;                     // 1 bit per base, LSB-first, 32 bases per word
    skh::DBuf<skh::ContigDesc> d_contigs;
    std::vector<skh::SeedTile> tiles;              // host tile list, ordered by (genome, contig, first)
    std::vector<uint32_t> genome_first_tile;       // n_genomes + 1: index of each genome's first tile (genomes without tiles: their successor's)
    std::vector<uint32_t> tile_cached_for;         // {mode} the tile list was built for
    skh::DBuf<skh::SeedTile> d_tiles;
    skh::DBuf<uint32_t> d_genome_first_tile;       // genome_first_tile on the device
    // filling state (pack_seed.hip genomes_begin / _append / _finish)
    bool open = false; uint64_t cap_units = 0, n_units = 0; uint32_t cap_contigs = 0, n_batches = 0;
    skh::DBuf<uint8_t> stage[2];                   // device staging of the batches' ASCII (host sources), alternating
    std::vector<std::shared_ptr<skh::DevEvent>> copied;   // one per batch: recorded behind the copy of its bases (skh_genomes_wait holds its own reference while it waits: _finish may clear the list meanwhile)
    std::mutex copied_mu;
};

// Device-resident Vec<Sketch>.  Index conventions: *_off are per-genome u64 offsets into the concatenated arrays.
struct skh_sketch_set {
    skh_ctx* ctx = nullptr;
    skh_sketch_params params{};
    uint32_t n_genomes = 0;
    // host metadata (one entry per genome unless noted)
    std::vector<uint64_t> pos_off, dist_off, mk_off, ctg_off, tab_off;   // n_genomes+1
    std::vector<uint32_t> n_buckets;               // home slots (buckets) of each genome's seed table
    std::vector<uint32_t> salt;                    // per genome: its table works on mix32(seed ^ salt) (common.h table_hash); 0 unless the seeds crowded the hash range
    std::vector<uint64_t> bmap_off;                // n_genomes+1: first 32-bit word of each genome's bucket-occupancy bitmap
    std::vector<uint64_t> ms_off;                  // n_genomes+1: first word of each genome's seed-list storage
    std::vector<uint32_t> ctg_len;                 // concatenated contig lengths
    std::vector<uint32_t> goff;                    // padded-coordinate start of every contig, n_contigs(g)+1 entries per genome at
                                                   // index ctg_off[g] + g (the last one = the genome's padded span)
    // A set with a genome of wide_span padded bases or more is WIDE: that genome's coordinates do not fit 31 bits.  The set then also keeps goff64 /
    // p_g64, the padded coordinates in 64 bits (of all its genomes), and for its wide genomes (wide_g) p_g holds the position's INDEX within the
    // genome << 1 | canonical instead: the seed tables and the join work on p_g exactly as they do otherwise (ascending index = ascending coordinate),
    // and the anchors of a pair with such a genome are translated to 64-bit coordinates before the chaining (chain.hip).  goff: exact for the others.
    bool wide = false, indexed = false;            // indexed: the wide genomes' p_g records have been made (sketch_build.hip)
    bool compact = false;                          // SKH_SKETCH_COMPACT: a set made to stay resident (a search database): 1.5 instead of 2 home slots per position, list
                                                   // storage cut to what the lists take -- 21 instead of 31 bytes per position, a slightly longer probe walk
    std::vector<uint8_t> wide_g;                   // per genome (wide sets only)
    std::vector<uint64_t> goff64;
    std::vector<uint64_t> total_len;
    std::vector<double> mean_ctg;
    std::vector<float> q10, q50, q90;
    // everything the chaining's pair descriptors take from one genome, gathered once per set (chain.hip genome_halves)
    struct GenomeHalf {
        const uint32_t *seed, *g, *rep, *ms, *bmap, *goff; const uint64_t* tab; const uint32_t* host_goff; uint32_t salt;
        const uint64_t *g64, *goff64, *host_goff64;       // a wide genome (else null)
        uint32_t n_pos, pos0, nbk, nctg, chunk_bound; uint64_t total_len; float q10, q50, q90;
        double score_markers, score_len;                  // switch_qr's two candidate scores (chain.rs:625-649)
    };
    mutable std::vector<GenomeHalf> halves;
    // what the per-pair host loop of a chaining call reads of a genome, in one cache line (the full halves take three): role decision, tile and chunk bounds
    struct HalfLite { double score_markers, score_len; uint64_t total_len; uint32_t n_pos, nbk, nctg, chunk_bound; uint8_t wide; };
    mutable std::vector<HalfLite> lite;
    // the device's copy of what its kernels need of the halves (chain_types.h GenomeDev; chain.hip dev_halves): uploaded with the first chaining call that uses the set,
    // on that context's stream; a context with another stream waits for the upload's event
    mutable skh::DBuf<char> d_halves; mutable bool d_halves_ok = false; mutable devStream_t d_halves_stream{}; mutable std::shared_ptr<skh::DevEvent> d_halves_ev;
    std::vector<uint32_t> rank;
    std::vector<std::string> names;                // optional file names (switch_qr tie-break)
    // device arrays
    skh::DBuf<uint32_t> p_seed, p_g;               // position order (contig, pos); p_g = padded coordinate << 1 | canonical (wide genomes: index in the genome << 1 | canonical)
    skh::DBuf<uint64_t> p_g64;                     // wide sets: padded coordinate << 1 | canonical, all genomes
    skh::DBuf<uint32_t> p_rep;                     // 1 bit per position (set-wide position index): its seed occurs more than 2500 / c times in its genome (chain.rs:674-676)
    // seed table (probe side), common.h: per genome n_buckets home slots in slices of TAB_SLICE, each followed by TAB_SLACK overflow slots;
    // slot = mix32(seed) << 32 | position (single seeds) / list reference / "repetitive"; TAB_EMPTY = free
    skh::DBuf<uint64_t> tab;                       // tab_off[g] .. (sketch_build.hip build_tables_kernel)
    skh::DBuf<uint32_t> ms;                        // list storage of the seeds with several positions: ms_off[g] + offset -> count, positions ascending
    skh::DBuf<uint32_t> bmap;                      // 1 bit per bucket: bucket non-empty (10 KB per 5 Mbp genome: staged in LDS by the join)
    skh::DBuf<uint64_t> markers;                   // sorted unique per genome
    skh::DBuf<uint32_t> d_goff;
    skh::DBuf<uint64_t> d_goff64;
    skh::DBuf<uint32_t> d_wide_g;
    // lazily built by the first two-set screen that uses this set as the reference side: its (marker, genome) incidences
    // sorted by marker (screen.hip).  The set is otherwise immutable; the mutex makes the one-time build safe when several
    // contexts share the set.
    mutable skh::DBuf<uint64_t> screen_keys;
    mutable skh::DBuf<uint32_t> screen_col_of, screen_genome_of;   // the triangle screen's column order (screen.hip), made behind the index at sketch time for sets of up to 16,384 genomes
    // The index made at sketch time is sorted on the context's second stream and the sketch call does NOT wait for it (round 5): its last passes run while the host
    // returns to its caller and comes back with the screen -- they used to be a 0.18 ms tail behind the table build in front of ~0.1 ms of host time.  What the
    // sort still works on is let go when the event behind it is done: by the screen that waited for it, by the context's next call, or with the set.
    // (PendingSort: the bucketed keys, the sort's counters and bounds -- or the radix sort's scratch -- and the event behind the sort; shared with the context that queued it, which lets go of the two buffers as soon
    //  as it sees the event done -- a resident database's shards are never screened themselves and would keep 16 bytes per marker for ever)
    mutable std::shared_ptr<skh::PendingSort> screen_sort;
    ~skh_sketch_set() { if (screen_sort) { try { screen_sort->ev.wait(); } catch (...) {} } }
    mutable std::mutex cache_mu;
    mutable std::mutex build_mu;                   // the (one-time) build of deferred seed tables: ensure_tables / skh_triangle's build beside its screen
    bool tables_built = false;                     // seed tables / filter / list storage exist (skh_sketch_genomes_ex may defer them: a rank of a distributed
                                                   // triangle indexes only the sketches it ends up chaining; ensure_tables builds them on first use)
    skh::DBuf<uint64_t> d_pos_off, d_dist_off, d_mk_off, d_ctg_off;
};

namespace skh {

// Collectives under the distributed triangle (dist.hip).  Two implementations: RCCL on device buffers (rccl_transport.hip) and caller-supplied
// host-memory collectives (device data staged through host buffers).  Every call returns when the data has arrived.
struct Transport {
    int rank = 0, world = 1;
    // Sizes of the communicator's previous distributed triangle (largest per-rank genome / contig / marker / candidate-pair / result-row counts): the next
    // call lays its gathers out by them and needs one collective where counts-then-payload would need two (dist.hip gather_records).
    uint64_t cap_n = 0, cap_c = 0, cap_m = 0, cap_pairs = 0, cap_rows = 0;
    // the key-range screen's cell gather runs on device buffers the communicator keeps (cap_cells + 2 words per rank: a head and the cells)
    uint64_t cap_cells = 0; DBuf<uint64_t> cells_send, cells_recv;
    std::vector<char> g_send, g_recv;                                               // host staging of gather_records (dist.hip)
    virtual ~Transport() {}
    // every rank contributes `bytes` bytes; recv gets world * bytes in rank order.  device: both buffers are device memory.
    virtual void all_gather(skh_ctx* ctx, const void* send, void* recv, size_t bytes, bool device) = 0;
    // send_cnt[r] bytes at send + send_off[r] go to rank r; recv_cnt[r] bytes from rank r land at recv + recv_off[r]
    virtual void all_to_all_v(skh_ctx* ctx, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                              const uint64_t* recv_off, bool device) = 0;
    // The same exchange on DEVICE buffers, asynchronously: _begin returns once the send buffer (complete on the context's stream at the call) has been handed
    // over; _end makes the received data visible to whatever is queued on the context's stream afterwards.  The count / offset arrays and both buffers stay
    // untouched in between.  One exchange at a time; no other collective of this communicator between the two calls.
    virtual void exchange_begin(skh_ctx* ctx, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                                const uint64_t* recv_off) = 0;
    virtual void exchange_end(skh_ctx* ctx) = 0;
    // microseconds the last exchange took from _begin to the arrival of the last byte, and how much of that the context's stream (or the calling thread)
    // spent waiting in _end; call when the context's stream is idle
    virtual void exchange_times(uint64_t* total_us, uint64_t* wait_us) = 0;
};
Transport* make_host_transport(const skh_host_collectives* hc, int rank, int world);    // dist.hip (the RCCL transport and its two entry points: rccl_transport.hip)

// SKH_TRACE=1: host wall-clock per stage of the host drivers on stderr (each mark synchronises the stream; diagnosis only)
struct StageTrace {
    skh_ctx* ctx; bool on, sync; std::chrono::steady_clock::time_point t;
    explicit StageTrace(skh_ctx* c) : ctx(c) { const char* v = getenv("SKH_TRACE"); on = v && (*v == '1' || *v == '2'); sync = on && *v == '1'; t = std::chrono::steady_clock::now(); }
    void mark(const char* what) {                        // SKH_TRACE=2: host time between marks without synchronising (where the host itself spends its time)
        if (!on) return;
        if (sync) dsync(ctx->stream);
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[skh trace] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// adds the stream time between construction and destruction to *dst (HIP events on the context's stream; synchronises at the end)
struct Stopwatch {   // wall-clock around stream-synchronous phases
    skh_ctx* ctx; float* dst; DevEvent e0, e1;
    Stopwatch(skh_ctx* c, float* d) : ctx(c), dst(d) { e0.record(ctx->stream); }
    ~Stopwatch() { try { e1.record(ctx->stream); e1.wait(); *dst += DevEvent::ms(e0, e1); } catch (...) {} }
};

// ---- scan.hip
void exclusive_scan_u32(skh_ctx* ctx, const uint32_t* d_in, uint64_t n, uint32_t* d_out /* n+1 entries */);
// up to eight regions of 32-bit words set to a value each, in one launch
struct FillRegions {
    uint32_t* p[8]; uint64_t words[8]; uint32_t value[8]; uint32_t blocks[8]; uint32_t n = 0;
    void add(void* ptr, uint64_t n_words, uint32_t v) { if (!n_words) return; if (n >= 8) throw skh::Error("FillRegions: more than eight regions"); p[n] = (uint32_t*)ptr; words[n] = n_words; value[n] = v; n++; }
};
void fill_regions(skh_ctx* ctx, FillRegions& fr);

// ---- sort (sort.hip): stable LSD radix sorts (rocPRIM) used while building sketches and the screen index
void sort_pairs_u32_u32(skh_ctx* ctx, uint32_t*& keys, uint32_t*& vals, uint64_t n, int end_bit);   // may redirect the pointers to the sorted arrays (arena)
void sort_keys_u64(skh_ctx* ctx, uint64_t* keys, uint64_t n, int end_bit, int begin_bit = 0);   // stable on bits [begin_bit, end_bit)
void sort_keys_u64_into(skh_ctx* ctx, uint64_t* keys, uint64_t* out, uint64_t n, int end_bit, DBuf<char>* tmp = nullptr);  // bits [0, end_bit); the result lands in `out`; tmp: the sort's scratch outside the arena (a sort that outlives the call)
uint64_t* sort_segments_u64(skh_ctx* ctx, uint64_t* keys, uint64_t n, uint32_t n_seg, const uint64_t* d_off, const uint64_t* h_off, int end_bit);   // every segment [off[s], off[s+1]) on bits [0, end_bit); returns the sorted array

// ---- pack_seed.hip
void genomes_begin(skh_ctx* ctx, skh_genome_set* gs, uint64_t max_bases, uint32_t max_contigs, uint32_t n_genomes);
void genomes_append(skh_ctx* ctx, skh_genome_set* gs, const uint8_t* bases, const uint64_t* contig_start, const uint64_t* contig_len, const uint32_t* contig_genome,
                    uint32_t n_contigs, int on_device, DevEvent* copied);
void genomes_finish(skh_ctx* ctx, skh_genome_set* gs);
struct SeedOutput {   // position-ordered raw seeding output for a whole genome set
    DBuf<uint32_t> seed, g; DBuf<uint64_t> markers_raw;       // g = padded coordinate << 1 | canonical (common.h CTG_PAD)
    DBuf<uint64_t> g64; bool wide = false;                    // a set with a genome beyond 31-bit coordinates: also g64 = the coordinates in 64 bits (g: their low words)
    std::vector<uint64_t> pos_off, mk_off;   // per genome, n_genomes+1
    bool tail_pending = false;               // the last kernel writing these arrays is still queued on the context's stream
};
void seed_genomes(skh_ctx* ctx, skh_genome_set* gs, const skh_sketch_params& sp, SeedOutput& out, bool async_tail = false, bool wide = false,
                  const std::function<void()>* meanwhile = nullptr);   // meanwhile: host work done once, behind the first seeding launch and in front of its read-back (the kernel runs for milliseconds)

// ---- sketch_build.hip
// needs p_seed, pos_off, contig tables (finalize_metadata) filled, and either p_g (pos == cc == null) or pos / cc = device
// arrays of (position in contig, contig << 1 | canonical) in position order, which are converted into p_g
void build_sketch_tables(skh_ctx* ctx, skh_sketch_set* ss, const uint32_t* pos, const uint32_t* cc);
struct TableBuild { uint32_t* d_back = nullptr; size_t n = 0; };                    // a table build that is queued but not yet waited for
TableBuild build_sketch_tables_begin(skh_ctx* ctx, skh_sketch_set* ss, const uint32_t* pos, const uint32_t* cc, skh::DevEvent* gate = nullptr);   // gate: build_tables_kernel starts behind this event (the marker sets' kernel on the other stream)
bool build_sketch_tables_finish(skh_ctx* ctx, skh_sketch_set* ss, TableBuild& tb);   // true: a salt changed or the list storage moved
void upload_set_offsets(skh_ctx* ctx, skh_sketch_set* ss);
void copy_segments(skh_ctx* ctx, const uint32_t* src, uint32_t* dst, const std::vector<uint64_t>& seg);   // dist.hip: segments of 32-bit words (src offset, dst offset, words) x n
void ensure_tables(skh_ctx* ctx, const skh_sketch_set* ss);                         // builds deferred tables (once; the set's mutex makes it safe across contexts)
// inverse of the padded-coordinate packing for export: fills device arrays pos / cc (either may be null) for entries [p0, p0+n)
void unpack_positions(skh_ctx* ctx, const skh_sketch_set* ss, uint64_t p0, uint64_t n, uint32_t* pos, uint32_t* cc);
struct ScreenKeysPlan {                                   // the sort's first half (buckets laid out, counted, scanned), kept for its second half
    bool valid = false, radix_only = false;
    uint32_t t_base = 0, shift = 0, nb = 0, rb = 0, gg = 0, n_ranges = 0, n_groups = 0, nbp = 0;
    uint32_t *hist = nullptr, *off = nullptr, *cursor = nullptr, *bounds = nullptr;
};
// The marker sets' one kernel launched ahead of everything else the marker build does, with an event behind it: the sketch call queues it BEFORE the seed tables and lets
// build_tables_kernel wait for that event.  Beside build_tables_kernel (seven workgroups of 22 KB of LDS per CU) marker_set_kernel's workgroups of 64 KB hardly ever find room:
// at 10,000 genomes it took 12.4 ms instead of 2.5 and the index sort ran alone behind the table build.  Beside slice_positions_kernel it is done first; the sort then runs
// beside the table build.  (1,000 genomes: the kernel ends before slice_positions_kernel does -- the gate is open when build_tables_kernel arrives.)
struct MarkerBuild { bool launched = false; uint64_t* d_ro = nullptr; uint32_t* d_uq = nullptr; DevEvent done; };
void build_markers_begin(skh_ctx* ctx, skh_sketch_set* ss, DBuf<uint64_t>& raw, const std::vector<uint64_t>& raw_off, MarkerBuild& mb);
void build_markers(skh_ctx* ctx, skh_sketch_set* ss, DBuf<uint64_t>& raw, const std::vector<uint64_t>& raw_off, ScreenKeysPlan* plan = nullptr, uint32_t* plan_max = nullptr, MarkerBuild* begun = nullptr);   // plan: also the first half of the screen's incidence sort (its scratch: the set's PendingSort), the largest bucket read back with the set sizes
void finalize_metadata(skh_sketch_set* ss);                                        // host-only: quantiles, means, padded contig starts

// ---- screen.hip
void reap_pending_sorts(skh_ctx* ctx);   // lets go of the scratch of index sorts that have finished
void prepare_screen_keys(skh_ctx* ctx, const skh_sketch_set* set, bool async = false, const ScreenKeysPlan* plan = nullptr, uint32_t plan_max = 0);   // async: the sort's last kernel is queued, the set's PendingSort event recorded behind it, not waited for
// screen_keys.hip: the incidence keys of the stretches [lo_g, lo_g + cnt_g) of every genome's (sorted) marker set, sorted by the marker's leading 16 bases, into out[0, n)
struct ScreenKeysIn { const uint64_t* markers; const uint64_t* mk_off; const uint64_t* range_lo /* null: mk_off[g] */; const uint32_t* range_cnt /* null: the whole set */; uint32_t ng; uint32_t is_query; };
bool sorted_screen_keys_fits(uint64_t n);
void screen_keys_count(skh_ctx* ctx, const ScreenKeysIn& in, uint64_t n_planned, uint64_t marker_lo, uint64_t marker_hi, PendingSort* own, uint32_t* d_max /* device: the largest bucket */, ScreenKeysPlan& plan);
void screen_keys_place(skh_ctx* ctx, const ScreenKeysIn& in, uint64_t n, const ScreenKeysPlan& plan, uint32_t h_max, uint64_t* out, PendingSort* own);
void sorted_screen_keys(skh_ctx* ctx, const ScreenKeysIn& in, uint64_t n, uint64_t marker_lo, uint64_t marker_hi /* 0: none; the range all stretches lie in */, uint64_t* out,
                        PendingSort* own /* null: scratch from the arena; else the sort's buffers, which outlive the call */);
void screen_pairs(skh_ctx* ctx, const skh_sketch_set* refs, const skh_sketch_set* queries, double identity, int rule,
                  int rescue_small, std::vector<uint32_t>& first, std::vector<uint32_t>& second,
                  uint32_t row_begin = 0, uint32_t row_end = 0xFFFFFFFFu);   // rows = queries (or refs when queries == NULL) restricted to [row_begin, row_end)

// the triangle screen cut by key range (the distributed triangle): the non-zero cells of the partial count matrix of part `part` of the markers' leading
// 16 bases, and the candidate pairs from the gathered cells of all parts (every rank gets the same list).  screen_parts_fit: the dense matrix is within the screen's budget.
bool screen_parts_fit(const skh_ctx* ctx, uint32_t n_genomes);
void screen_partial_cells(skh_ctx* ctx, const skh_sketch_set* S, uint32_t part, uint32_t n_parts, std::vector<uint64_t>& cells);   // a cell: i << 43 | j << 22 | count
void screen_from_cells(skh_ctx* ctx, const skh_sketch_set* S, const uint64_t* cells, uint64_t n_cells, double identity, int rescue_small,
                       std::vector<uint32_t>& first, std::vector<uint32_t>& second);
// the device forms (the distributed triangle: the cells never visit the host): cells left in the arena; cells read from blocks [count, -, cells...] of block_words words
void screen_partial_cells_dev(skh_ctx* ctx, const skh_sketch_set* S, uint32_t part, uint32_t n_parts, uint64_t** d_cells, uint64_t* n_cells,
                              uint64_t all_from = 0, uint64_t all_below = 0);   // n_parts == 1: S holds markers of [all_from, all_below) only (0: no bound) -- what the sort's buckets are laid over
uint64_t screen_part_bound(uint32_t r, uint32_t n_parts);
void screen_marker_parts(skh_ctx* ctx, const skh_sketch_set* S, uint32_t n_parts, std::vector<uint64_t>& lo, std::vector<uint32_t>& cnt);
void screen_from_cells_dev(skh_ctx* ctx, const skh_sketch_set* S, const uint64_t* d_blocks, uint32_t n_blocks, uint64_t block_words, uint64_t max_cells, double identity, int rescue_small,
                           std::vector<uint32_t>& first, std::vector<uint32_t>& second);

// ---- dist.hip
void assign_pairs(uint32_t n_genomes, const std::vector<uint32_t>& pi, const std::vector<uint32_t>& pj, const std::vector<uint64_t>& weight, const std::vector<int>& holder,
                  int world, std::vector<uint8_t>& owner, std::vector<uint64_t>& units_of, std::vector<uint64_t>& load);
void triangle_distributed(skh_ctx* ctx, Transport& T, const skh_sketch_set* local, double identity, int rescue_small, const skh_map_params& mp, uint32_t flags,
                          std::vector<uint32_t>& out_i, std::vector<uint32_t>& out_j, std::vector<skh_ani_result>& out_res, uint64_t* n_chained, skh_dist_stats* stats);
void comm_selftest(skh_ctx* ctx, Transport& T);

// ---- chain.hip
void prepare_halves(skh_ctx* ctx, const skh_sketch_set* S, bool ahead = false, bool again = false);   // per-genome tables (host + device) ahead of the first chaining call; ahead: while the set's table build is still running; again: drop what was made ahead
// chain_seeds for a list of pairs; pair p takes its reference from Rsets[pair_rset[p]] and its query from Qsets[pair_qset[p]] (null set-index array: set 0)
void chain_pairs(skh_ctx* ctx, const skh_sketch_set* const* Rsets, uint32_t n_rsets, const uint32_t* pair_rset, const skh_sketch_set* const* Qsets, uint32_t n_qsets,
                 const uint32_t* pair_qset, const uint32_t* pair_ref, const uint32_t* pair_query, uint64_t n_pairs, const skh_map_params& mp, skh_ani_result* out,
                 skh_chain_stats* stats, bool tie_by_rank = false,   // tie_by_rank: switch_qr's tie (chain.rs:20-22) goes by genome_rank even when both sets carry file names
                 const std::function<void()>* tables_pending = nullptr);   // the sets' table build is still queued (build_sketch_tables_begin): the pair descriptors are made
                                                                           // while it runs, then this is called -- it must finish the build -- and the pairs are chained

}  // namespace skh
