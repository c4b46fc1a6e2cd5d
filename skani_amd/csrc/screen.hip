// screen.hip -- marker k-mer screen as a batched set-intersection prefilter.
//
// Replaces screen.rs:190-210 (kmer_to_sketch_from_refs, the marker -> genome-ids inverted index), :148-189
// (screen_refs), :39-77 (screen_refs_indices) and :84-142 (check_markers_quickly).
// GPU formulation: all (marker, genome) incidences are sorted by marker (the inverted index becomes runs of
// equal markers; screen_keys.hip); every incidence adds 1 to count[row][col] for each co-occurring genome on the other side; a
// second pass applies the reference's exact cut-off rule per cell and compacts the passing pairs in
// (row, col) order.  Counts are exact integers, so the pass set is identical to the reference's.
#include <algorithm>

#include "internal.h"
#include <cmath>

namespace skh {

constexpr int ID_BITS = SCREEN_ID_BITS;            // genome id field inside the sort key (common.h screen_key; a marker is 42 bits)
constexpr uint64_t ID_MASK = SCREEN_ID_MASK;

__device__ __forceinline__ uint32_t seg_of64(const uint64_t* off, uint32_t n_seg, uint64_t i) {
    uint32_t lo = 0, hi = n_seg;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// key = (marker's low 10 bits << 22 | is_query << 21 | genome) << 32 | marker >> 10.  The incidence lists are sorted by the keys' LOW 32 bits only --
// the marker's leading 16 bases: half the radix passes of a full sort; the incidences of one marker then sit somewhere inside their prefix group,
// in no particular order, and the count kernels walk the group and compare whole markers.  Distinct markers that share a prefix are rare (tens
// of millions of markers over 2^32 prefixes).  The sorted field sits in the low bits because rocPRIM 4.2's radix_sort_keys with begin_bit > 0
// returns unsorted output for 1,200 .. 1,000,000 keys (tools/exp/rocprim_bits.hip; sorting bits [0, 32) is fine at every size) -- the radix sort is what
// rounds 1-4 sorted the lists with and what screen_keys.hip still hands a list to when one of its buckets does not fit the LDS.
constexpr int SCREEN_SORT_BITS = 32;
constexpr uint64_t SCREEN_MARKER_MASK = ~(((1ull << (ID_BITS + 1)) - 1ull) << 32);         // everything but is_query and genome
__device__ __forceinline__ uint32_t skey_genome(uint64_t key) { return (uint32_t)(key >> 32) & (uint32_t)ID_MASK; }
__device__ __forceinline__ uint32_t skey_prefix(uint64_t key) { return (uint32_t)key; }
__device__ __forceinline__ bool skey_same_marker(uint64_t a, uint64_t b) { return ((a ^ b) & SCREEN_MARKER_MASK) == 0; }
__global__ __launch_bounds__(256) void screen_keys_kernel(const uint64_t* markers, const uint64_t* mk_off, uint32_t ng, uint64_t n,
                                                          uint32_t is_query, uint64_t* keys) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = screen_key(markers[i], is_query, seg_of64(mk_off, ng, i));
}

// triangle: incidence (m, a) pairs with every later incidence (m, b) of its prefix group  ->  count[min(a, b) - row0][max(a, b)]
// The counters are kept once per XCD (planes): a device-scope atomic leaves the XCD's L2 for the fabric (~20 M of them per step at 24 G/s),
// while an atomic on memory that only this XCD touches during the kernel can stay in its L2 (workgroup scope = no sc1 write-through; the L2
// itself is what makes it atomic among the XCD's CUs).  The plane is chosen by the hardware's XCC id, not by the block number.  The
// threshold kernel adds the planes up.
__device__ __forceinline__ void count_local(uint32_t* p) { atomic_inc_xcd_local(p); }   // (dev.h: xcc_id(), the XCD-local increment)
// Self-test of the per-XCD planes, run once per context before they are used: 512 workgroups add into 64 counters of "their" plane with the same
// XCD-local atomic; the planes must add up to exactly the number of increments.  The HIP memory model does not promise that workgroup-scope
// atomics of different workgroups see each other -- on gfx950 they do, because an XCD performs them in its one L2 -- so the planes are only used
// after this hardware / driver / partition mode has shown that it behaves that way (otherwise: one device-scope plane).
__global__ __launch_bounds__(256) void screen_planes_selftest_kernel(uint32_t* cnt, uint32_t n_planes) {
    uint32_t* plane = cnt + (size_t)(xcc_id() % n_planes) * 64;
    for (uint32_t r = 0; r < 4; r++) count_local(&plane[(threadIdx.x + r) & 63u]);
}

// FIRST (one plane only): the increment that finds a cell at zero also counts the cell in its row's number of non-zero cells (row_nz): the emission of the cells then
// needs no counting pass over the matrix
template <bool FIRST>
__global__ __launch_bounds__(256) void screen_count_tri_kernel(const uint64_t* keys, uint64_t n, uint32_t row0, uint32_t rows, uint32_t ncols,
                                                               uint32_t* cnt, uint32_t n_planes, uint64_t plane, uint32_t* row_nz) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint64_t key = keys[e];
    const uint32_t a = skey_genome(key), prefix = skey_prefix(key);
    uint32_t* mine = cnt + (n_planes > 1 ? (uint64_t)(xcc_id() % n_planes) * plane : 0ull);
    for (uint64_t f = e + 1; f < n; f++) {
        const uint64_t k2 = keys[f];
        if (skey_prefix(k2) != prefix) break;
        if (!skey_same_marker(k2, key)) continue;
        const uint32_t b = skey_genome(k2), lo = a < b ? a : b, hi = a < b ? b : a;
        if (lo < row0 || lo >= row0 + rows) continue;
        uint32_t* cell = mine + (uint64_t)(lo - row0) * ncols + hi;
        // (FIRST asks for the increment's old value.  Measured at 10,000 genomes: the parts of a world of 2 / 4 / 8 take 3.9 / 2.2 / 1.34 ms against 4.4 / 2.6 / 1.8 ms with plain
        // increments + zeroing + a counting pass over the matrix; one part over all 50 M keys 7.5 ms (clade order) / 6.9 ms (shuffled) against 8.0 / 7.2 ms.)
        if (FIRST) { if (atomicAdd(cell, 1u) == 0u) atomicAdd(&row_nz[lo - row0], 1u); }
        else if (n_planes > 1) count_local(cell); else atomicAdd(cell, 1u);
    }
}

// The same counting, a ROW of the matrix per instruction (round 6).  What the count costs is requests to the L2's atomic units, and a request is a 64-byte LINE, not a
// word: tools/exp/atomic_rates.hip (profiles/r06_atomic_rates.md) measures 27 G increments/s when every lane of an instruction hits a line of its own -- whatever the
// number of hot words, the scope, the planes -- and 395 G/s when the 16 lanes of a quarter wave hit the 16 words of one line.  In the kernel above lane = incidence e and
// step s pairs it with incidence e + s: the lanes of one instruction are in different rows, one line each.  Here the lanes of a marker's group walk the group TOGETHER
// from its first incidence: at step i every lane of the group meets member i, and the lanes whose genome is the larger of the two add to cell (member i, own genome) --
// one row, columns as far apart as the group's genomes are.  Genomes that share markers tend to sit next to each other in a collection sorted by file name (the
// clades of the synthetic collections do): a group of m then costs ~m line requests instead of m (m - 1) / 2.  Where they do not, it is one line per increment as before.
// Every unordered pair of a group is met once (by its larger genome), so the counts are the same integers.  A workgroup stages its COUNT_TILE keys and COUNT_HALO keys
// on either side in LDS; the part of a group beyond that (a marker shared by hundreds of genomes) is read from global memory.
constexpr uint32_t COUNT_TILE = 1024, COUNT_HALO = 256;
template <bool FIRST, uint32_t TILE = COUNT_TILE>                                    // (tiles of 256 for the column order's small sample: four times the workgroups)
__global__ __launch_bounds__(256) void screen_count_tri_rows_kernel(const uint64_t* keys, uint64_t n, uint32_t row0, uint32_t rows, uint32_t ncols,
                                                                    uint32_t* cnt, uint32_t n_planes, uint64_t plane, uint32_t* row_nz, const uint32_t* col_of /* null: columns = genomes */) {
    __shared__ uint64_t sk[TILE + 2 * COUNT_HALO];
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint64_t st_lo = base >= COUNT_HALO ? base - COUNT_HALO : 0, st_hi = n - base < TILE + COUNT_HALO ? n : base + TILE + COUNT_HALO;
    for (uint32_t x = threadIdx.x; x < (uint32_t)(st_hi - st_lo); x += blockDim.x) sk[x] = keys[st_lo + x];
    __syncthreads();
    auto key_at = [&](uint64_t p) { return (p >= st_lo && p < st_hi) ? sk[p - st_lo] : keys[p]; };
    uint32_t* mine = cnt + (n_planes > 1 ? (uint64_t)(xcc_id() % n_planes) * plane : 0ull);
    const uint64_t own_hi = n - base < TILE ? n : base + TILE;
    for (uint64_t e = base + threadIdx.x; e < own_hi; e += blockDim.x) {
        const uint64_t key = key_at(e);
        const uint32_t b = skey_genome(key), prefix = skey_prefix(key);
        const uint32_t bcol = col_of ? col_of[b] : b;
        uint64_t gs = e;                                                               // the prefix group's first incidence
        while (gs > 0 && skey_prefix(key_at(gs - 1)) == prefix) gs--;
        for (uint64_t p = gs; p < n; p++) {
            const uint64_t k2 = key_at(p);
            if (skey_prefix(k2) != prefix) break;
            const uint32_t a = skey_genome(k2);
            if (a >= b || !skey_same_marker(k2, key) || a < row0 || a >= row0 + rows) continue;   // (a == b: the incidence itself)
            uint32_t* cell = mine + (uint64_t)(a - row0) * ncols + bcol;
            if (FIRST) { if (atomicAdd(cell, 1u) == 0u) atomicAdd(&row_nz[a - row0], 1u); }
            else if (n_planes > 1) count_local(cell); else atomicAdd(cell, 1u);
        }
    }
}

// two sets, separately sorted key arrays: a query incidence (m, q) finds m's prefix group in the refs' keys by binary search.  The refs'
// sorted keys are cached in the sketch set (a database is screened many times; its index is built once).
__global__ __launch_bounds__(256) void screen_count_qr2_kernel(const uint64_t* qkeys, uint64_t nq, const uint64_t* rkeys, uint64_t nr, uint32_t row0, uint32_t rows,
                                                               uint32_t ncols, uint32_t* cnt) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nq) return;
    const uint64_t key = qkeys[e];
    const uint32_t q = skey_genome(key);
    if (q < row0 || q >= row0 + rows) return;
    const uint32_t prefix = skey_prefix(key);
    uint64_t lo = 0, hi = nr;                                                        // first ref key of the prefix group
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (skey_prefix(rkeys[mid]) < prefix) lo = mid + 1; else hi = mid; }
    uint32_t* row = cnt + (uint64_t)(q - row0) * ncols;
    for (uint64_t f = lo; f < nr; f++) {
        const uint64_t k2 = rkeys[f];
        if (skey_prefix(k2) != prefix) break;
        if (skey_same_marker(k2, key)) atomicAdd(&row[skey_genome(k2)], 1u);
    }
}

// ---- the COLUMN ORDER of the triangle's count matrix (round 6).  The row-per-instruction kernel above pays one line request per increment again when the genomes of a
// group are far apart in the collection -- related genomes under unrelated file names.  The matrix is the screen's own scratch, so its columns may stand in any order:
//   1. the head of the sorted list (~64 incidences per genome) is counted into the (zeroed) matrix as it stands, columns = genomes;
//   2. a workgroup per row reads its cells, puts them back to zero, and unites the row's genome with every genome it shares at least `thr` of those sampled markers
//      with -- a lock-free union-find in global memory (parents only ever decrease; ~19 links per genome).  A single shared marker does NOT tie two genomes: unrelated
//      genomes share a marker here and there (5,200 such pairs among the 1,000 genomes of the synthetic collection, tools/exp/cross_clade_cells.py; in real collections
//      far more), and clusters tied by single markers swallow each other;
//   3. genomes are ordered by (smallest genome of their cluster, own number): a collection that IS in clade order keeps its order.
// Cell (row a, genome b) then sits in column col_of[b]; the rule kernels walk the columns and name genome genome_of[column]; the host orders every row's few columns
// again (order_rows_columns).  Which genomes end up next to each other changes no count: the pass set is the same.
__global__ __launch_bounds__(256) void colorder_init_kernel(uint32_t* parent, uint32_t n) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) parent[g] = g;
}
__device__ __forceinline__ uint32_t uf_root(uint32_t* parent, uint32_t x) {          // (a stale parent is an earlier ancestor: still in the cluster, still leads to the root)
    for (;;) { const uint32_t q = load_past_l1(&parent[x]); if (q == x) return x; x = q; }
}
__device__ __forceinline__ void uf_unite(uint32_t* parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_root(parent, a); b = uf_root(parent, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }                             // a > b: hook a under b
        const uint32_t old = atomicMin(&parent[a], b);
        if (old == a) return;                                                         // a was a root: done
        a = old;                                                                      // a had been hooked meanwhile: its former parent and b still have to meet
    }
}
__global__ __launch_bounds__(256) void colorder_links_kernel(uint32_t* cnt, uint32_t N, uint32_t thr, uint32_t* parent) {
    const uint32_t row = blockIdx.x;
    uint32_t* crow = cnt + (uint64_t)row * N;
    for (uint32_t col = ((row + 1) & ~63u) + threadIdx.x; col < N; col += blockDim.x) {
        if (col <= row) continue;
        const uint32_t c = crow[col];
        if (!c) continue;
        crow[col] = 0;                                                                // the matrix leaves as it came: all zero
        if (c >= thr) uf_unite(parent, row, col);
    }
}
__global__ __launch_bounds__(256) void colorder_label_kernel(const uint32_t* parent, uint32_t n, uint64_t* lab) {   // (its own launch: every parent is final)
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint32_t r = g; for (uint32_t q; (q = parent[r]) != r;) r = q;
    lab[g] = (uint64_t)r << 32 | g;
}
constexpr uint32_t COLORDER_LDS_MAX = 4096;
// up to COLORDER_LDS_MAX genomes: every workgroup puts all labels into its LDS (its own launch: every parent is final) and each of its threads counts the labels below its genome's
__global__ __launch_bounds__(256) void colorder_rank_kernel(const uint32_t* parent, uint32_t n, uint32_t* col_of, uint32_t* genome_of) {
    __shared__ uint64_t s[COLORDER_LDS_MAX];
    for (uint32_t g = threadIdx.x; g < n; g += blockDim.x) { uint32_t r = g; for (uint32_t q; (q = parent[r]) != r;) r = q; s[g] = (uint64_t)r << 32 | g; }
    __syncthreads();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint64_t mine = s[g]; uint32_t rank = 0;
    for (uint32_t x = 0; x < n; x++) rank += s[x] < mine ? 1u : 0u;                     // (labels are distinct: the genome is part of them)
    col_of[g] = rank; genome_of[rank] = g;
}
__global__ __launch_bounds__(256) void colorder_place_kernel(const uint64_t* sorted, uint32_t n, uint32_t* col_of, uint32_t* genome_of) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint32_t g = (uint32_t)sorted[x];
    genome_of[x] = g; col_of[g] = x;
}
// The order's working arrays: the context's arena for an order made inside a screen call; buffers of the set's PendingSort for the one made at sketch time on the second
// stream, which nobody waits for (the arena is reset when the sketch call returns).  `cnt`: N x N zeroed words of scratch, left zeroed.
struct ColOrderWork { uint32_t* parent; uint64_t* lab; uint64_t* sorted; DBuf<char>* sort_tmp; };
static bool column_order_wanted(const skh_ctx* ctx, uint32_t N, uint64_t n_keys) { return ctx->tune.screen_count_rows && ctx->tune.screen_col_order && N >= 2 && n_keys >= 2; }
static void queue_column_order(skh_ctx* ctx, const uint64_t* keys, uint64_t n_keys, uint32_t N, uint32_t* cnt, const ColOrderWork& w, uint32_t* col_of, uint32_t* genome_of) {
    const uint64_t n_sample = std::min<uint64_t>(n_keys, std::max<uint64_t>((uint64_t)64 * N, (uint64_t)1 << 15));
    const uint32_t thr = (uint32_t)std::max<uint64_t>(3, n_sample / N / 32);         // ~3 % of a genome's sampled markers (the screen's own cut-off is at 0.9 %: pairs that matter share far more)
    SKH_LAUNCH(colorder_init_kernel, (N + 255) / 256, 256, 0, ctx->stream, w.parent, N);
    SKH_LAUNCH((screen_count_tri_rows_kernel<false, 256>), (unsigned)((n_sample + 255) / 256), 256, 0, ctx->stream, keys, n_sample, 0u, N, N, cnt, 1u, (uint64_t)N * N, (uint32_t*)nullptr,
               (const uint32_t*)nullptr);
    SKH_LAUNCH(colorder_links_kernel, N, 256, 0, ctx->stream, cnt, N, thr, w.parent);
    if (N <= COLORDER_LDS_MAX) SKH_LAUNCH(colorder_rank_kernel, (N + 255) / 256, 256, 0, ctx->stream, (const uint32_t*)w.parent, N, col_of, genome_of);
    else {
        SKH_LAUNCH(colorder_label_kernel, (N + 255) / 256, 256, 0, ctx->stream, (const uint32_t*)w.parent, N, w.lab);
        sort_keys_u64_into(ctx, w.lab, w.sorted, N, 64, w.sort_tmp);
        SKH_LAUNCH(colorder_place_kernel, (N + 255) / 256, 256, 0, ctx->stream, (const uint64_t*)w.sorted, N, col_of, genome_of);
    }
    check_launch("screen column order");
}
// inside a screen call: col_of / genome_of in the context's arena (null when the order is switched off: columns = genomes)
static void make_column_order(skh_ctx* ctx, const uint64_t* keys, uint64_t n_keys, uint32_t N, uint32_t* cnt, uint32_t** col_of, uint32_t** genome_of) {
    *col_of = nullptr; *genome_of = nullptr;
    if (!column_order_wanted(ctx, N, n_keys)) return;
    const ColOrderWork w{ctx->arena.get<uint32_t>(N), ctx->arena.get<uint64_t>(N), ctx->arena.get<uint64_t>(N), nullptr};
    uint32_t* co = ctx->arena.get<uint32_t>(N); uint32_t* go = ctx->arena.get<uint32_t>(N);
    queue_column_order(ctx, keys, n_keys, N, cnt, w, co, go);
    *col_of = co; *genome_of = go;
}
// at sketch time, behind the index sort on the same stream: the order is cached in the set, the matrix it needs for a moment belongs to the PendingSort.  Up to 16,384
// genomes (a 1 GB matrix); larger collections make the order inside their screen call, whose own matrix is there anyway.
constexpr uint32_t COLORDER_AHEAD_MAX = 16384;
static void queue_column_order_ahead(skh_ctx* ctx, const skh_sketch_set* set, PendingSort* ps) {
    const uint32_t N = set->n_genomes; const uint64_t n_keys = set->screen_keys.n;
    if (!column_order_wanted(ctx, N, n_keys) || N > COLORDER_AHEAD_MAX || set->compact) return;   // (compact sets are the shards of a resident database: searched, never screened against themselves)
    try { ps->mat.alloc((size_t)N * N); ps->work.alloc((size_t)3 * N); set->screen_col_of.alloc(N); set->screen_genome_of.alloc(N); }
    catch (...) {                                                                    // no room for the sample's matrix (up to 1 GB): the screen call makes the order in its own matrix, or counts in collection order
        ps->mat.release(); ps->work.release(); set->screen_col_of.release(); set->screen_genome_of.release();
        return;
    }
    dzero(ps->mat.p, (size_t)N * N * 4, ctx->stream);
    const ColOrderWork w{(uint32_t*)ps->work.p, ps->work.p + N, ps->work.p + 2 * (size_t)N, &ps->sort_tmp};
    queue_column_order(ctx, set->screen_keys.p, n_keys, N, ps->mat.p, w, set->screen_col_of.p, set->screen_genome_of.p);
}
// the candidates of rows [from, end) of `first` come out of the rule kernels in column order: every row's genomes ascending again (triangle.rs:90 walks them that way)
static void order_rows_columns(const std::vector<uint32_t>& first, std::vector<uint32_t>& second, size_t from) {
    for (size_t x = from; x < first.size();) {
        size_t y = x + 1; while (y < first.size() && first[y] == first[x]) y++;
        if (y - x > 1 && !std::is_sorted(second.begin() + x, second.begin() + y)) std::sort(second.begin() + x, second.begin() + y);   // (a collection in clade order: nothing to do)
        x = y;
    }
}

struct ScreenRule { double cutoff; int rule; int rescue_small; int triangle; };
constexpr int SCREEN_RULE_NONZERO = 100;                                          // internal: "the cell has a count" (the partial counts of a key range, screen_partial_cells)

__device__ __forceinline__ bool cell_passes(const ScreenRule& sr, uint32_t count, uint64_t m_row, uint64_t m_col, uint32_t row, uint32_t col) {
    if (sr.triangle && col <= row) return false;                                 // triangle.rs:90
    if (sr.rule == SCREEN_RULE_NONZERO) return count != 0;
    const uint64_t mn = m_row < m_col ? m_row : m_col;
    if (sr.rule == SKH_SCREEN_QUICK) {                                           // screen.rs:84-142
        if (mn < SCREEN_MIN_KMERS && sr.rescue_small) return true;
        if (mn == 0) return sr.rescue_small != 0;
        uint64_t ratio = (uint64_t)(sr.cutoff * (double)mn);
        if (ratio == 0) ratio = 1;
        return (uint64_t)count >= ratio;
    }
    if (sr.rule == SKH_SCREEN_REFS && m_row < SCREEN_MIN_KMERS && sr.rescue_small) return true;   // screen.rs:158-160
    if (count == 0) return false;                                                // only refs present in the count map can pass
    uint64_t thr = (uint64_t)(sr.cutoff * (double)mn);                           // screen.rs:176-187 / :64-75
    if (thr < 1) thr = 1;
    return (uint64_t)count > thr;
}

// one workgroup per row: pass 0 counts passing cells, pass 1 writes them in column order
__global__ __launch_bounds__(256) void screen_threshold_kernel(const uint32_t* cnt, uint32_t n_planes, uint64_t plane, uint32_t row0, uint32_t ncols, ScreenRule sr,
                                                               const uint64_t* mk_off_rows, const uint64_t* mk_off_cols, int pass,
                                                               uint32_t* row_cnt, const uint32_t* row_off, uint32_t* out_first, uint32_t* out_second, uint32_t* out_count /* may be null */,
                                                               const uint32_t* genome_of /* null: column c is genome c; else the column order of make_column_order */) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t running;
    const uint32_t r = blockIdx.x, row = row0 + r;
    const uint64_t m_row = mk_off_rows[row + 1] - mk_off_rows[row];
    const uint32_t* crow = cnt + (uint64_t)r * ncols;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const uint32_t base_out = pass ? row_off[r] : 0;
    if (genome_of) {
        // columns in an order of their own: the host puts every row's candidates in order anyway (order_rows_columns), so a row is read without a barrier per 256 columns --
        // the threads count (pass 0) or take their slots from a counter in LDS (pass 1)
        uint32_t mine = 0;
        const bool zero_may_pass = sr.rescue_small && (sr.rule == SKH_SCREEN_QUICK || (sr.rule == SKH_SCREEN_REFS && m_row < SCREEN_MIN_KMERS));   // (cell_passes: the rescued small genomes)
        for (uint32_t at = threadIdx.x; at < ncols; at += blockDim.x) {
            uint32_t count = 0;
            for (uint32_t pl = 0; pl < n_planes; pl++) count += crow[(uint64_t)pl * plane + at];
            if (!count && !zero_may_pass) continue;                                    // (most of a row: no look-ups for it)
            const uint32_t col = genome_of[at];
            if (!cell_passes(sr, count, m_row, mk_off_cols[col + 1] - mk_off_cols[col], row, col)) continue;
            if (!pass) { mine++; continue; }
            const uint32_t o = base_out + atomicAdd(&running, 1u);
            out_first[o] = row; out_second[o] = col; if (out_count) out_count[o] = count;
        }
        if (!pass) { if (mine) atomicAdd(&running, mine); __syncthreads(); if (threadIdx.x == 0) row_cnt[r] = running; }
        return;
    }
    for (uint32_t c0 = 0; c0 < ncols; c0 += blockDim.x) {
        const uint32_t col = c0 + threadIdx.x;
        bool ok = false; uint32_t count = 0;
        if (col < ncols) {
            for (uint32_t pl = 0; pl < n_planes; pl++) count += crow[(uint64_t)pl * plane + col];
            ok = cell_passes(sr, count, m_row, mk_off_cols[col + 1] - mk_off_cols[col], row, col);
        }
        // workgroup exclusive scan of the flags
        uint32_t incl = wave_incl_scan(ok ? 1u : 0u);
        const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
        if (l == 63) lds[w] = incl;
        __syncthreads();
        uint32_t before = 0, tot = 0;
        for (uint32_t i = 0; i < (blockDim.x >> 6); i++) { uint32_t t = lds[i]; if (i < w) before += t; tot += t; }
        const uint32_t run = running;
        if (pass && ok) { const uint32_t o = base_out + run + before + incl - 1; out_first[o] = row; out_second[o] = col; if (out_count) out_count[o] = count; }
        __syncthreads();
        if (threadIdx.x == 0) running = run + tot;
        __syncthreads();
    }
    if (!pass && threadIdx.x == 0) row_cnt[r] = running;
}

static double powi21(double a) {   // f64::powi(x, 21) lowers to compiler-rt __powidf2: square-and-multiply from the low bit
    int b = (int)K_MARKER; double r = 1;
    while (true) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
    return r;
}

// fills the set's cache of sorted (marker, genome) incidences on the context's current stream (no-op when present)
// `premade`: the set's keys in (genome, marker) order, already written by the marker build (scratch: sorted from there into the cache)
void reap_pending_sorts(skh_ctx* ctx) {
    auto& v = ctx->pending_sorts;
    for (size_t x = 0; x < v.size();) { if (v[x]->ev.done()) { v[x]->release(); v[x] = v.back(); v.pop_back(); } else x++; }
}
void prepare_screen_keys(skh_ctx* ctx, const skh_sketch_set* set, bool async, const ScreenKeysPlan* plan, uint32_t plan_max) {
    const uint32_t ng = set->n_genomes;
    if (!ng || ng > ID_MASK) { set->screen_sort.reset(); return; }
    const uint64_t MR = set->mk_off[ng];
    std::lock_guard<std::mutex> lk(set->cache_mu);
    if (set->screen_keys.n == MR && MR) return;
    if (!sorted_screen_keys_fits(MR)) return;                                        // (the screen makes them itself then, the long way)
    set->screen_keys.alloc(MR ? MR : 1);
    if (MR) {
        // async: the last kernel is queued and the caller returns; whoever uses the index waits for the event on its stream.  The sort's scratch is the set's own then.
        if (async && !set->screen_sort) set->screen_sort.reset(new PendingSort());
        const ScreenKeysIn in{set->markers.p, set->d_mk_off.p, nullptr, nullptr, ng, 0u};
        if (plan && plan->valid && async) screen_keys_place(ctx, in, MR, *plan, plan_max, set->screen_keys.p, set->screen_sort.get());   // (counted by the marker build already)
        else sorted_screen_keys(ctx, in, MR, 0, 0, set->screen_keys.p, async ? set->screen_sort.get() : nullptr);
        if (async) { queue_column_order_ahead(ctx, set, set->screen_sort.get()); set->screen_sort->ev.record(ctx->stream); ctx->pending_sorts.push_back(set->screen_sort); return; }
    }
    dsync(ctx->stream);
    set->screen_sort.reset();
}

void screen_pairs(skh_ctx* ctx, const skh_sketch_set* refs, const skh_sketch_set* queries, double identity, int rule, int rescue_small,
                  std::vector<uint32_t>& first, std::vector<uint32_t>& second, uint32_t row_begin, uint32_t row_end) {
    first.clear(); second.clear();
    if (identity == 0.) identity = 0.80;                                          // triangle.rs:34-42, SEARCH_ANI_CUTOFF_DEFAULT
    const bool tri = queries == nullptr;
    const skh_sketch_set* rowset = tri ? refs : queries;
    const uint32_t nrows = rowset->n_genomes, ncols = refs->n_genomes;
    if (nrows == 0 || ncols == 0) return;
    StageTrace tr(ctx);
    if (ncols > ID_MASK || nrows > ID_MASK) throw Error("more than 2M genomes in one screen call");
    const uint64_t MR = refs->mk_off[ncols], MQ = tri ? 0 : queries->mk_off[nrows], M = MR + MQ;
    bool keys_pending = false;
    const uint64_t* keys = nullptr;      // triangle: the one sorted incidence list; two sets: the queries' list
    const uint64_t* rkeys = nullptr;     // two sets: the refs' sorted incidence list (cached in the set)
    auto make_keys = [&](const skh_sketch_set* set, uint32_t n_genomes, uint64_t n, uint32_t is_query, uint64_t* out) {
        if (!n) return;
        if (sorted_screen_keys_fits(n)) { sorted_screen_keys(ctx, ScreenKeysIn{set->markers.p, set->d_mk_off.p, nullptr, nullptr, n_genomes, is_query}, n, 0, 0, out, nullptr); return; }
        uint64_t* raw = ctx->arena.get<uint64_t>(n);                                   // beyond 2^32 incidences: keys in set order, a device-wide radix sort
        SKH_LAUNCH(screen_keys_kernel, (unsigned)((n + 255) / 256), 256, 0, ctx->stream, (const uint64_t*)set->markers.p, (const uint64_t*)set->d_mk_off.p, n_genomes, n,
                   is_query, raw);
        check_launch("screen_keys");
        sort_keys_u64_into(ctx, raw, out, n, SCREEN_SORT_BITS);   // nothing leans on the order inside a prefix group
    };
    // the set's (marker, genome) incidences sorted by marker are cached in the set: built on first use, or ahead of time by prepare_screen_keys
    // (skh_sketch_genomes does that on its second stream while the seed tables are built, which takes the sort off the triangle's critical path)
    {
        std::lock_guard<std::mutex> lk(refs->cache_mu);
        if (refs->screen_keys.n != MR || MR == 0) { refs->screen_keys.alloc(MR ? MR : 1); make_keys(refs, ncols, MR, 0u, refs->screen_keys.p); dsync(ctx->stream); }
        else if (refs->screen_sort) { refs->screen_sort->ev.make_wait(ctx->stream); keys_pending = true; }   // made at sketch time, possibly still being sorted on that context's second stream
    }
    if (tri) keys = refs->screen_keys.p;
    else {
        rkeys = refs->screen_keys.p;
        uint64_t* k = ctx->arena.get<uint64_t>(MQ ? MQ : 1);
        make_keys(queries, nrows, MQ, 1u, k); keys = k;
    }
    tr.mark("screen: keys + sort");
    ScreenRule sr{powi21(identity), rule, rescue_small, tri ? 1 : 0};
    // row blocking keeps the dense count matrix within a fixed budget
    const uint64_t budget_cells = ctx->tune.screen_cells;   // u32 counters per row block (default 8 GiB)
    uint32_t rows_per = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(1, std::min(row_end, nrows) - std::min(row_begin, nrows)), std::max<uint64_t>(1, budget_cells / ncols));
    // triangle: one plane of counters per XCD while that stays small (8 x 4 MB for 1000 genomes); the two-set screen of a large database
    // keeps the single device-scope plane
    const uint64_t plane = (uint64_t)rows_per * ncols;
    if (tri && ctx->tune.screen_planes > 1 && !ctx->screen_planes_checked) {
        const uint32_t np = std::min<uint32_t>(ctx->tune.screen_planes, 8u), blocks = 512;
        uint32_t* d = ctx->arena.get<uint32_t>((size_t)np * 64); dzero(d, (size_t)np * 64 * 4, ctx->stream);
        SKH_LAUNCH(screen_planes_selftest_kernel, blocks, 256, 0, ctx->stream, d, np);
        check_launch("screen_planes_selftest");
        std::vector<uint32_t> h((size_t)np * 64);
        d2h(h.data(), d, h.size() * 4, ctx->stream);
        uint64_t sum = 0; for (uint32_t v : h) sum += v;
        if (sum != (uint64_t)blocks * 256 * 4) ctx->tune.screen_planes = 1;          // increments were lost: XCD-local atomics are not safe here
        ctx->screen_planes_checked = true;
    }
    const uint32_t want_planes = std::min<uint32_t>(std::max<uint32_t>(ctx->tune.screen_planes, 1u), 8u);
    const uint32_t n_planes = (tri && plane * want_planes <= (64ull << 20)) ? want_planes : 1u;
    uint32_t* cnt = ctx->arena.get<uint32_t>(plane * n_planes);
    uint32_t* row_cnt = ctx->arena.get<uint32_t>(rows_per); uint32_t* row_off = ctx->arena.get<uint32_t>(rows_per + 1);
    row_end = std::min(row_end, nrows);
    uint32_t* col_of = nullptr; uint32_t* genome_of = nullptr;                      // triangle: the count matrix's columns in an order of their own (when the whole matrix is one row block: it is the order's scratch)
    bool zeroed = false;
    if (tri && M && ctx->tune.screen_count_rows && ctx->tune.screen_col_order) {
        if (refs->screen_col_of.n == ncols && refs->screen_genome_of.n == ncols) { col_of = refs->screen_col_of.p; genome_of = refs->screen_genome_of.p; }   // made at sketch time, behind the index sort (whose event this stream has waited for)
        else if (row_begin == 0 && rows_per >= nrows) { dzero(cnt, plane * n_planes * 4, ctx->stream); make_column_order(ctx, keys, MR, ncols, cnt, &col_of, &genome_of); zeroed = col_of != nullptr; }
    }                                                                                // (zeroed: the order's scratch was the whole matrix, zeroed in front of it and left zeroed)
    for (uint32_t row0 = row_begin; row0 < row_end; row0 += rows_per) {
        const uint32_t rows = std::min(rows_per, row_end - row0);
        if (!(zeroed && row0 == row_begin)) dzero(cnt, plane * n_planes * 4, ctx->stream);
        if (M) {
            if (tri && ctx->tune.screen_count_rows) SKH_LAUNCH(screen_count_tri_rows_kernel<false>, (unsigned)((MR + COUNT_TILE - 1) / COUNT_TILE), 256, 0, ctx->stream, keys, MR, row0, rows, ncols, cnt, n_planes, plane, (uint32_t*)nullptr, (const uint32_t*)col_of);
            else if (tri) SKH_LAUNCH(screen_count_tri_kernel<false>, (unsigned)((MR + 255) / 256), 256, 0, ctx->stream, keys, MR, row0, rows, ncols, cnt, n_planes, plane, (uint32_t*)nullptr);
            else if (MQ && MR) SKH_LAUNCH(screen_count_qr2_kernel, (unsigned)((MQ + 255) / 256), 256, 0, ctx->stream, keys, MQ, rkeys, MR, row0, rows, ncols, cnt);
            check_launch("screen_count");
        }
        SKH_LAUNCH(screen_threshold_kernel, rows, 256, 0, ctx->stream, (const uint32_t*)cnt, n_planes, plane, row0, ncols, sr, (const uint64_t*)rowset->d_mk_off.p,
                   (const uint64_t*)refs->d_mk_off.p, 0, row_cnt, (const uint32_t*)row_off, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)genome_of);
        check_launch("screen_threshold0");
        exclusive_scan_u32(ctx, row_cnt, rows, row_off);
        uint32_t total = 0;
        d2h(&total, row_off + rows, 4, ctx->stream);
        if (total) {
            uint32_t* of = ctx->arena.get<uint32_t>(2 * (size_t)total); uint32_t* os = of + total;   // (side by side: one read-back)
            SKH_LAUNCH(screen_threshold_kernel, rows, 256, 0, ctx->stream, (const uint32_t*)cnt, n_planes, plane, row0, ncols, sr, (const uint64_t*)rowset->d_mk_off.p,
                       (const uint64_t*)refs->d_mk_off.p, 1, row_cnt, (const uint32_t*)row_off, of, os, (uint32_t*)nullptr, (const uint32_t*)genome_of);
            check_launch("screen_threshold1");
            std::vector<uint32_t> both(2 * (size_t)total);
            d2h(both.data(), of, both.size() * 4, ctx->stream);
            const size_t from = first.size();
            first.insert(first.end(), both.begin(), both.begin() + total); second.insert(second.end(), both.begin() + total, both.end());
            if (genome_of) order_rows_columns(first, second, from);
        }
    }
    dsync(ctx->stream);
    if (keys_pending) {                                                              // this stream waited for the sort and is idle now: the sort is over, its scratch can go
        std::lock_guard<std::mutex> lk(refs->cache_mu);
        if (refs->screen_sort) { refs->screen_sort->release(); refs->screen_sort.reset(); }
    }
    tr.mark("screen: count + threshold");
}

// ---- the triangle screen cut by KEY RANGE (the distributed triangle, dist.hip).  Every rank holds all marker sets; cutting the triangle by rows made every
// rank sort and walk ALL (marker, genome) incidences and only spared it the increments of the other ranks' rows -- the screen did not get faster with the
// number of GPUs.  Cut by the marker's leading 16 bases instead, a rank sorts and walks a W-th of the incidences (a marker's incidences all lie in one
// range) and gets PARTIAL counts for all cells; the non-zero cells (the pairs that share a marker of the range: a few per related pair) are gathered, every
// rank adds them up in a dense matrix and applies the rule to all rows itself -- the same candidate list on every rank, no list to gather.
// A genome's markers are sorted, and the key's sorted field (marker >> 10) rises with the marker: the markers of a range are a stretch of every genome's set.
__global__ __launch_bounds__(256) void screen_part_ranges_kernel(const uint64_t* markers, const uint64_t* mk_off, uint32_t ng, uint64_t lo_marker, uint64_t hi_marker /* 0 = no upper bound */,
                                                                 uint64_t* range_lo, uint32_t* range_cnt) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const uint64_t a = mk_off[g], b = mk_off[g + 1];
    auto first_ge = [&](uint64_t v) { uint64_t lo = a, hi = b; while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (markers[mid] < v) lo = mid + 1; else hi = mid; } return lo; };
    const uint64_t x = first_ge(lo_marker), y = hi_marker ? first_ge(hi_marker) : b;
    range_lo[g] = x; range_cnt[g] = (uint32_t)(y - x);
}
// a cell on the wire: i << 43 | j << 22 | count (i, j < 2^21; a count beyond 2^22 - 1 -- genomes with millions of markers -- saturates, which no threshold notices)
constexpr uint32_t CELL_COUNT_MAX = (1u << 22) - 1u;
__global__ __launch_bounds__(256) void screen_pack_cells_kernel(const uint32_t* ci, const uint32_t* cj, const uint32_t* cc, uint32_t n, uint64_t* out) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = ((uint64_t)ci[e] << 43) | ((uint64_t)cj[e] << 22) | (cc[e] < CELL_COUNT_MAX ? cc[e] : CELL_COUNT_MAX);
}
// cells in blocks of `block_words` words (one per part, as the gather of the distributed triangle leaves them): word 0 = the block's number of cells, word 1 unused,
// the cells from word 2 on; blockIdx.y = the block
__global__ __launch_bounds__(256) void screen_add_cells_kernel(const uint64_t* blocks, uint64_t block_words, uint32_t ng, uint32_t* cnt, uint32_t* bad) {
    const uint64_t* blk = blocks + (uint64_t)blockIdx.y * block_words;
    const uint64_t n = blk[0];
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || e + 2 >= block_words) return;
    const uint64_t c = blk[2 + e]; const uint32_t i = (uint32_t)(c >> 43), j = (uint32_t)(c >> 22) & 0x1FFFFFu;
    if (i >= ng || j >= ng) { atomicAdd(bad, 1u); return; }
    uint32_t* cell = &cnt[(uint64_t)i * ng + j];
    const uint32_t add = (uint32_t)c & CELL_COUNT_MAX, old = atomicAdd(cell, add);
    if (old + add < old) atomicMax(cell, 0xFFFFFFFFu);                                // (saturate instead of wrapping)
}

// ---- the gathered cells WITHOUT the dense matrix (round 5).  Every part's cells come out of threshold_rows in (i, j) order, so row i of the count matrix is a stretch of
// every block: a workgroup per row finds the stretches (one binary search pair per block), adds them up in an LDS row of N counters and puts that row through the
// rule -- the same two passes as screen_threshold_kernel (count, then write in column order), reading LDS where that one reads N x N words of HBM.  What the dense form
// cost at 10,000 genomes: 400 MB zeroed, 400 MB read twice -- the same on every rank, whatever the world.
__global__ __launch_bounds__(256) void screen_cells_check_kernel(const uint64_t* blocks, uint64_t block_words, uint32_t ng, uint32_t* flags /* [0] a genome beyond the set, [1] a block out of row order */) {
    const uint64_t* blk = blocks + (uint64_t)blockIdx.y * block_words;
    const uint64_t n = blk[0];
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || e + 2 >= block_words) return;
    const uint64_t c = blk[2 + e]; const uint32_t i = (uint32_t)(c >> 43), j = (uint32_t)(c >> 22) & 0x1FFFFFu;
    if (i >= ng || j >= ng) atomicOr(&flags[0], 1u);
    if (e + 1 < n && e + 3 < block_words && (uint32_t)(blk[3 + e] >> 43) < i) atomicOr(&flags[1], 1u);
}
__global__ __launch_bounds__(256) void screen_rows_from_cells_kernel(const uint64_t* blocks, uint64_t block_words, uint32_t n_blocks, uint32_t ng, ScreenRule sr, const uint64_t* mk_off, int pass,
                                                                     uint32_t* row_cnt, const uint32_t* row_off, uint32_t* out_first, uint32_t* out_second) {
    SKH_DYN_SMEM(smem);
    uint32_t* cnt = (uint32_t*)smem;                                                  // ng counters: this row of the count matrix
    __shared__ uint32_t seg_lo[256], seg_hi[256];
    __shared__ uint32_t lds[16];
    __shared__ uint32_t running, any;
    const uint32_t row = blockIdx.x;
    if (threadIdx.x == 0) { running = 0; any = 0; }
    for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) {                   // this row's stretch of block b: [first cell with i >= row, first with i > row)
        const uint64_t* blk = blocks + (uint64_t)b * block_words;
        const uint32_t n = (uint32_t)blk[0];
        auto first_ge = [&](uint32_t v) { uint32_t lo = 0, hi = n; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)(blk[2 + mid] >> 43) < v) lo = mid + 1; else hi = mid; } return lo; };
        const uint32_t a = first_ge(row), z = first_ge(row + 1);
        seg_lo[b] = a; seg_hi[b] = z;
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < n_blocks; b += blockDim.x) if (seg_hi[b] > seg_lo[b]) any = 1;   // (benign race: every writer stores 1)
    __syncthreads();
    const uint64_t m_row = mk_off[row + 1] - mk_off[row];
    const bool all_pass = sr.rule == SKH_SCREEN_REFS && m_row < SCREEN_MIN_KMERS && sr.rescue_small;   // screen.rs:158-160: every later genome passes, counted or not
    if (!any && !all_pass) { if (!pass && threadIdx.x == 0) row_cnt[row] = 0; return; }               // (a row nobody shares a marker with: most rows of a collection of unrelated clades)
    for (uint32_t c = threadIdx.x; c < ng; c += blockDim.x) cnt[c] = 0;
    __syncthreads();
    for (uint32_t b = 0; b < n_blocks; b++) {
        const uint64_t* blk = blocks + (uint64_t)b * block_words + 2;
        for (uint32_t e = seg_lo[b] + threadIdx.x; e < seg_hi[b]; e += blockDim.x) {
            const uint64_t c = blk[e]; const uint32_t j = (uint32_t)(c >> 22) & 0x1FFFFFu;
            if (j < ng) atomicAdd(&cnt[j], (uint32_t)c & CELL_COUNT_MAX);              // (at most 255 parts x 2^22: no wrap)
        }
    }
    __syncthreads();
    const uint32_t base_out = pass ? row_off[row] : 0;
    for (uint32_t c0 = (row + 1) & ~255u; c0 < ng; c0 += blockDim.x) {                // triangle.rs:90: columns beyond the row only
        const uint32_t col = c0 + threadIdx.x;
        const bool ok = col < ng && cell_passes(sr, cnt[col < ng ? col : 0], m_row, mk_off[(col < ng ? col : 0) + 1] - mk_off[col < ng ? col : 0], row, col);
        const uint32_t incl = wave_incl_scan(ok ? 1u : 0u);
        const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
        if (l == 63) lds[w] = incl;
        __syncthreads();
        uint32_t before = 0, tot = 0;
        for (uint32_t i = 0; i < (blockDim.x >> 6); i++) { const uint32_t t = lds[i]; if (i < w) before += t; tot += t; }
        const uint32_t run = running;
        if (pass && ok) { const uint32_t o = base_out + run + before + incl - 1; out_first[o] = row; out_second[o] = col; }
        __syncthreads();
        if (threadIdx.x == 0) running = run + tot;
        __syncthreads();
    }
    if (!pass && threadIdx.x == 0) row_cnt[row] = running;
}

// one workgroup per row of a single-plane count matrix: the row's non-zero cells (columns beyond the row) go out as packed words in column order at row_off[row] --
// the numbers of non-zero cells per row are known from the counting (row_nz) -- and every cell read is put back to ZERO: the matrix leaves the call as it entered it
__global__ __launch_bounds__(256) void screen_emit_cells_kernel(uint32_t* cnt, uint32_t ncols, const uint32_t* row_nz, const uint32_t* row_off, uint64_t* out, const uint32_t* genome_of /* null: column = genome */) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t running;
    const uint32_t row = blockIdx.x;
    if (row_nz[row] == 0) return;
    uint32_t* crow = cnt + (uint64_t)row * ncols;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    const uint32_t base_out = row_off[row];
    for (uint32_t c0 = genome_of ? 0u : (row + 1) & ~255u; c0 < ncols; c0 += blockDim.x) {   // (in genome order the row's cells lie beyond the diagonal; in column order anywhere)
        const uint32_t at = c0 + threadIdx.x;
        const uint32_t count = (at < ncols && (genome_of || at > row)) ? crow[at] : 0u;
        const bool ok = count != 0;
        const uint32_t col = ok && genome_of ? genome_of[at] : at;
        if (ok) crow[at] = 0;
        const uint32_t incl = wave_incl_scan(ok ? 1u : 0u);
        const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
        if (l == 63) lds[w] = incl;
        __syncthreads();
        uint32_t before = 0, tot = 0;
        for (uint32_t i = 0; i < (blockDim.x >> 6); i++) { const uint32_t t = lds[i]; if (i < w) before += t; tot += t; }
        const uint32_t run = running;
        if (ok) out[base_out + run + before + incl - 1] = ((uint64_t)row << 43) | ((uint64_t)col << 22) | (count < CELL_COUNT_MAX ? count : CELL_COUNT_MAX);
        __syncthreads();
        if (threadIdx.x == 0) running = run + tot;
        __syncthreads();
    }
}

// rows [0, rows) of a dense count matrix through the rule: the passing (row, col[, count]) cells in (row, col) order, on the host
static void threshold_rows(skh_ctx* ctx, const uint32_t* cnt, uint32_t n_planes, uint64_t plane, uint32_t row0, uint32_t rows, uint32_t ncols, const ScreenRule& sr, const uint64_t* d_mk_rows,
                           const uint64_t* d_mk_cols, std::vector<uint32_t>& first, std::vector<uint32_t>& second, uint64_t** d_cells = nullptr /* packed (row, col, count) words left in the arena instead of first / second */,
                           uint64_t* n_cells = nullptr, const uint32_t* genome_of = nullptr /* the matrix's column order (make_column_order) */) {
    const bool cells = d_cells != nullptr;
    if (cells) { *d_cells = nullptr; *n_cells = 0; }
    uint32_t* row_cnt = ctx->arena.get<uint32_t>(rows); uint32_t* row_off = ctx->arena.get<uint32_t>(rows + 1);
    SKH_LAUNCH(screen_threshold_kernel, rows, 256, 0, ctx->stream, cnt, n_planes, plane, row0, ncols, sr, d_mk_rows, d_mk_cols, 0, row_cnt, (const uint32_t*)row_off, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, genome_of);
    check_launch("screen_threshold0");
    exclusive_scan_u32(ctx, row_cnt, rows, row_off);
    uint32_t total = 0;
    d2h(&total, row_off + rows, 4, ctx->stream);
    if (!total) return;
    uint32_t* of = ctx->arena.get<uint32_t>(2 * (size_t)total); uint32_t* os = of + total; uint32_t* oc = cells ? ctx->arena.get<uint32_t>(total) : nullptr;
    SKH_LAUNCH(screen_threshold_kernel, rows, 256, 0, ctx->stream, cnt, n_planes, plane, row0, ncols, sr, d_mk_rows, d_mk_cols, 1, row_cnt, (const uint32_t*)row_off, of, os, oc, genome_of);
    check_launch("screen_threshold1");
    if (cells) {
        uint64_t* packed = ctx->arena.get<uint64_t>(total);
        SKH_LAUNCH(screen_pack_cells_kernel, (total + 255) / 256, 256, 0, ctx->stream, (const uint32_t*)of, (const uint32_t*)os, (const uint32_t*)oc, total, packed);
        check_launch("screen_pack_cells");
        *d_cells = packed; *n_cells = total;
        return;
    }
    std::vector<uint32_t> both(2 * (size_t)total);
    d2h(both.data(), of, both.size() * 4, ctx->stream);
    const size_t from = first.size();
    first.insert(first.end(), both.begin(), both.begin() + total); second.insert(second.end(), both.begin() + total, both.end());
    if (genome_of) order_rows_columns(first, second, from);
}

// the smallest marker of part r of n_parts (0 for r = 0 and for r >= n_parts: no bound)
uint64_t screen_part_bound(uint32_t r, uint32_t n_parts) {
    if (r == 0 || r >= n_parts) return 0ull;
    const double x = 1.0 - std::sqrt(1.0 - (double)r / (double)n_parts);
    return std::max<uint64_t>((uint64_t)(x * 4294967296.0), 1ull) << 10;
}
// per genome of S and part r of n_parts: where the genome's markers of that part begin (an index into S->markers) and how many they are -- the stretches a rank of the
// distributed triangle sends to the rank that screens part r (dist.hip)
__global__ __launch_bounds__(256) void screen_all_part_ranges_kernel(const uint64_t* markers, const uint64_t* mk_off, uint32_t ng, const uint64_t* bounds /* n_parts + 1; 0 = none */, uint32_t n_parts,
                                                                     uint64_t* lo_out, uint32_t* cnt_out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (g >= ng) return;
    const uint64_t a = mk_off[g], b = mk_off[g + 1];
    auto first_ge = [&](uint64_t v) { uint64_t lo = a, hi = b; while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (markers[mid] < v) lo = mid + 1; else hi = mid; } return lo; };
    const uint64_t x = first_ge(bounds[r]), y = bounds[r + 1] ? first_ge(bounds[r + 1]) : b;
    lo_out[(uint64_t)g * n_parts + r] = x; cnt_out[(uint64_t)g * n_parts + r] = (uint32_t)(y - x);
}
void screen_marker_parts(skh_ctx* ctx, const skh_sketch_set* S, uint32_t n_parts, std::vector<uint64_t>& lo, std::vector<uint32_t>& cnt) {
    const uint32_t N = S->n_genomes;
    lo.assign((size_t)N * n_parts, 0); cnt.assign((size_t)N * n_parts, 0);
    if (!N || !S->mk_off[N]) return;
    std::vector<uint64_t> bounds(n_parts + 1);
    for (uint32_t r = 0; r <= n_parts; r++) bounds[r] = screen_part_bound(r, n_parts);
    uint64_t* d_b = ctx->arena.get<uint64_t>(n_parts + 1); uint64_t* d_lo = ctx->arena.get<uint64_t>(lo.size()); uint32_t* d_cnt = ctx->arena.get<uint32_t>(cnt.size());
    h2d(d_b, bounds.data(), bounds.size() * 8, ctx->stream);
    SKH_LAUNCH(screen_all_part_ranges_kernel, dim3((N + 255) / 256, n_parts), 256, 0, ctx->stream, (const uint64_t*)S->markers.p, (const uint64_t*)S->d_mk_off.p, N, (const uint64_t*)d_b, n_parts, d_lo, d_cnt);
    check_launch("screen_all_part_ranges");
    d2h(lo.data(), d_lo, lo.size() * 8, ctx->stream); d2h(cnt.data(), d_cnt, cnt.size() * 4, ctx->stream);
}

bool screen_parts_fit(const skh_ctx* ctx, uint32_t n_genomes) { return n_genomes && (uint64_t)n_genomes * n_genomes <= ctx->tune.screen_cells && n_genomes <= ID_MASK; }

// the non-zero cells of the triangle's count matrix over the markers whose leading 16 bases fall into part `part` of `n_parts`
void screen_partial_cells(skh_ctx* ctx, const skh_sketch_set* S, uint32_t part, uint32_t n_parts, std::vector<uint64_t>& cells) {
    cells.clear();
    uint64_t* d = nullptr; uint64_t n = 0;
    screen_partial_cells_dev(ctx, S, part, n_parts, &d, &n);
    cells.resize(n);
    if (n) d2h(cells.data(), d, n * 8, ctx->stream);
}
// the same, the cells left in the context's arena (valid until its next reset)
void screen_partial_cells_dev(skh_ctx* ctx, const skh_sketch_set* S, uint32_t part, uint32_t n_parts, uint64_t** d_cells, uint64_t* n_cells, uint64_t all_from, uint64_t all_below) {
    *d_cells = nullptr; *n_cells = 0;
    std::vector<uint32_t> none_a, none_b;
    const uint32_t N = S->n_genomes;
    if (!screen_parts_fit(ctx, N)) throw Error("screen_partial_cells: the count matrix does not fit");
    if (!S->mk_off[N]) return;
    StageTrace tr(ctx);
    // the smallest marker of part r (its sorted field is marker >> 10, 32 bits).  A marker is the smaller of a 21-mer and its reverse complement, so a fraction 1 - (1 - x)^2 of
    // them lies below x of the range: the parts are cut at the quantiles of that distribution, not at equal widths (two equal halves would hold 75 % and 25 % of the keys).
    // Every rank computes the same bounds: IEEE division and square root are exactly rounded.
    auto bound = [&](uint32_t r) -> uint64_t { return screen_part_bound(r, n_parts); };
    uint64_t* range_lo = ctx->arena.get<uint64_t>(N); uint32_t* range_cnt = ctx->arena.get<uint32_t>(N); uint32_t* part_off = ctx->arena.get<uint32_t>(N + 1);
    SKH_LAUNCH(screen_part_ranges_kernel, (N + 255) / 256, 256, 0, ctx->stream, (const uint64_t*)S->markers.p, (const uint64_t*)S->d_mk_off.p, N, bound(part), bound(part + 1), range_lo, range_cnt);
    check_launch("screen_part_ranges");
    tr.mark("screen part: ranges kernel");
    exclusive_scan_u32(ctx, range_cnt, N, part_off);
    tr.mark("screen part: scan");
    uint32_t n = 0;
    d2h(&n, part_off + N, 4, ctx->stream);
    tr.mark("screen part: d2h n");
    if (!n) return;
    uint64_t* keys = ctx->arena.get<uint64_t>(n);
    sorted_screen_keys(ctx, ScreenKeysIn{S->markers.p, S->d_mk_off.p, range_lo, range_cnt, N, 0u}, n, n_parts == 1 ? all_from : bound(part), n_parts == 1 ? all_below : bound(part + 1), keys, nullptr);
    tr.mark("screen part: keys, sorted");
    const uint64_t plane = (uint64_t)N * N;
    uint32_t* col_of = nullptr; uint32_t* genome_of = nullptr;                      // (each part orders the columns by what ITS keys tie together: the cells name genomes)
    // The order is made inside this call (a pass over half the matrix for the links, a sort of N labels, the cells then emitted from whole rows): it pays for itself from ~1,000
    // incidences per genome on -- config 4's 10,000 genomes: the parts of a world of 1 / 2 / 4 / 8 took 5.3 / 3.5 / 2.1 / 1.56 ms with it and 6.0 / 3.9 / 2.2 / 1.34 without.
    const bool order_pays = ctx->tune.screen_col_order > 1 || (uint64_t)n >= (uint64_t)1024 * N;
    // one plane of counters per XCD while that stays small (as in screen_pairs; the planes have passed their self-test there or are not used)
    const uint32_t want_planes = std::min<uint32_t>(std::max<uint32_t>(ctx->tune.screen_planes, 1u), 8u);
    const uint32_t n_planes = (ctx->screen_planes_checked && plane * want_planes <= (64ull << 20)) ? want_planes : 1u;
    if (n_planes == 1) {
        // a large collection (at 10,000 genomes the matrix is 400 MB): the context's own matrix, zero between calls; counting notes every row's number of non-zero cells,
        // emitting puts the cells back to zero.  (Before: the matrix zeroed, counted, then read twice -- 1.2 GB of traffic for ~160,000 cells on every rank.)
        if (ctx->part_cnt.n < plane || !ctx->part_cnt_clean) {
            if (ctx->part_cnt.n < plane) ctx->part_cnt.alloc(plane);
            dzero(ctx->part_cnt.p, ctx->part_cnt.n * 4, ctx->stream);
        }
        ctx->part_cnt_clean = false;
        if (ctx->tune.screen_count_rows && order_pays) make_column_order(ctx, keys, n, N, ctx->part_cnt.p, &col_of, &genome_of);
        uint32_t* row_nz = ctx->arena.get<uint32_t>(N); uint32_t* row_off = ctx->arena.get<uint32_t>(N + 1);
        dzero(row_nz, (size_t)N * 4, ctx->stream);
        if (ctx->tune.screen_count_rows) SKH_LAUNCH(screen_count_tri_rows_kernel<true>, (n + COUNT_TILE - 1) / COUNT_TILE, 256, 0, ctx->stream, (const uint64_t*)keys, (uint64_t)n, 0u, N, N, ctx->part_cnt.p, 1u, plane, row_nz, (const uint32_t*)col_of);
        else SKH_LAUNCH(screen_count_tri_kernel<true>, (n + 255) / 256, 256, 0, ctx->stream, (const uint64_t*)keys, (uint64_t)n, 0u, N, N, ctx->part_cnt.p, 1u, plane, row_nz);
        check_launch("screen_count(part)");
        tr.mark("screen part: count (first touch)");
        exclusive_scan_u32(ctx, row_nz, N, row_off);
        uint32_t total = 0;
        d2h(&total, row_off + N, 4, ctx->stream);
        uint64_t* packed = ctx->arena.get<uint64_t>(total ? total : 1);
        SKH_LAUNCH(screen_emit_cells_kernel, N, 256, 0, ctx->stream, ctx->part_cnt.p, N, (const uint32_t*)row_nz, (const uint32_t*)row_off, packed, (const uint32_t*)genome_of);
        check_launch("screen_emit_cells");
        dsync(ctx->stream);
        tr.mark("screen part: emit");
        ctx->part_cnt_clean = true;
        *d_cells = total ? packed : nullptr; *n_cells = total;
        return;
    }
    uint32_t* cnt = ctx->arena.get<uint32_t>(plane * n_planes);
    dzero(cnt, plane * n_planes * 4, ctx->stream);
    if (ctx->tune.screen_count_rows && order_pays) make_column_order(ctx, keys, n, N, cnt, &col_of, &genome_of);
    if (ctx->tune.screen_count_rows) SKH_LAUNCH(screen_count_tri_rows_kernel<false>, (n + COUNT_TILE - 1) / COUNT_TILE, 256, 0, ctx->stream, (const uint64_t*)keys, (uint64_t)n, 0u, N, N, cnt, n_planes, plane, (uint32_t*)nullptr, (const uint32_t*)col_of);
    else SKH_LAUNCH(screen_count_tri_kernel<false>, (n + 255) / 256, 256, 0, ctx->stream, (const uint64_t*)keys, (uint64_t)n, 0u, N, N, cnt, n_planes, plane, (uint32_t*)nullptr);
    check_launch("screen_count(part)");
    tr.mark("screen part: zero + count");
    const ScreenRule sr{0., SCREEN_RULE_NONZERO, 0, 1};
    threshold_rows(ctx, cnt, n_planes, plane, 0, N, N, sr, S->d_mk_off.p, S->d_mk_off.p, none_a, none_b, d_cells, n_cells, genome_of);
    dsync(ctx->stream);
    tr.mark("screen part: threshold + pack");
}

// the triangle's candidate pairs from the gathered cells of all parts: counts added up in a dense matrix, every row through the rule (triangle.rs:71-90 with screen_refs)
void screen_from_cells(skh_ctx* ctx, const skh_sketch_set* S, const uint64_t* cells, uint64_t n_cells, double identity, int rescue_small,
                       std::vector<uint32_t>& first, std::vector<uint32_t>& second) {
    first.clear(); second.clear();
    if (!screen_parts_fit(ctx, S->n_genomes)) throw Error("screen_from_cells: the count matrix does not fit");
    uint64_t* d = ctx->arena.get<uint64_t>(n_cells + 2);                              // one block: its head, the cells
    const uint64_t head[2] = {n_cells, 0};
    h2d(d, head, 16, ctx->stream);
    if (n_cells) h2d(d + 2, cells, n_cells * 8, ctx->stream);                         // (the caller's array is pageable: through the pinned ring, dev.h)
    screen_from_cells_dev(ctx, S, d, 1, n_cells + 2, n_cells, identity, rescue_small, first, second);
}
// the cells in device memory, in `n_blocks` blocks of `block_words` words (screen_add_cells_kernel); max_cells: the largest block's number of cells
void screen_from_cells_dev(skh_ctx* ctx, const skh_sketch_set* S, const uint64_t* d_blocks, uint32_t n_blocks, uint64_t block_words, uint64_t max_cells, double identity, int rescue_small,
                           std::vector<uint32_t>& first, std::vector<uint32_t>& second) {
    first.clear(); second.clear();
    if (identity == 0.) identity = 0.80;
    const uint32_t N = S->n_genomes;
    if (!screen_parts_fit(ctx, N)) throw Error("screen_from_cells: the count matrix does not fit");
    if (max_cells + 2 > block_words || max_cells > 0x7FFFFFFFull * 256) throw Error("screen_from_cells: bad block layout");
    const ScreenRule sr{powi21(identity), SKH_SCREEN_REFS, rescue_small, 1};
    // The sparse form: blocks in row order (what screen_partial_cells_dev makes), a row of counters that fits the LDS, at most 256 blocks.  Anything else -- cells a
    // caller put together some other way, a collection beyond ~40,000 genomes -- takes the dense matrix below.
    const size_t row_bytes = (size_t)N * 4;
    if (!ctx->tune.screen_cells_dense && n_blocks <= 256 && row_bytes <= ((size_t)150 << 10)) {
        uint32_t* flags = ctx->arena.get<uint32_t>(2); dzero(flags, 8, ctx->stream);
        if (max_cells && n_blocks) {
            SKH_LAUNCH(screen_cells_check_kernel, dim3((unsigned)((max_cells + 255) / 256), n_blocks), 256, 0, ctx->stream, d_blocks, block_words, N, flags);
            check_launch("screen_cells_check");
        }
        uint32_t h_flags[2] = {0, 0}; d2h(h_flags, flags, 8, ctx->stream);
        if (h_flags[0]) throw std::invalid_argument("screen_from_cells: a cell names a genome beyond the set");
        if (!h_flags[1]) {
            uint32_t* row_cnt = ctx->arena.get<uint32_t>(N); uint32_t* row_off = ctx->arena.get<uint32_t>(N + 1);
            kernel_allow_lds(screen_rows_from_cells_kernel, row_bytes);              // (a row of more than 16,384 genomes is beyond the 64 KB a launch may ask for unannounced, dev.h)
            SKH_LAUNCH(screen_rows_from_cells_kernel, N, 256, row_bytes, ctx->stream, d_blocks, block_words, n_blocks, N, sr, (const uint64_t*)S->d_mk_off.p, 0, row_cnt, (const uint32_t*)row_off,
                       (uint32_t*)nullptr, (uint32_t*)nullptr);
            check_launch("screen_rows_from_cells0");
            exclusive_scan_u32(ctx, row_cnt, N, row_off);
            uint32_t total = 0;
            d2h(&total, row_off + N, 4, ctx->stream);
            if (total) {
                uint32_t* of = ctx->arena.get<uint32_t>(2 * (size_t)total); uint32_t* os = of + total;
                SKH_LAUNCH(screen_rows_from_cells_kernel, N, 256, row_bytes, ctx->stream, d_blocks, block_words, n_blocks, N, sr, (const uint64_t*)S->d_mk_off.p, 1, row_cnt, (const uint32_t*)row_off, of, os);
                check_launch("screen_rows_from_cells1");
                std::vector<uint32_t> both(2 * (size_t)total);
                d2h(both.data(), of, both.size() * 4, ctx->stream);
                first.assign(both.begin(), both.begin() + total); second.assign(both.begin() + total, both.end());
            }
            dsync(ctx->stream);
            return;
        }
    }
    const uint64_t plane = (uint64_t)N * N;
    uint32_t* cnt = ctx->arena.get<uint32_t>(plane);
    dzero(cnt, plane * 4, ctx->stream);
    if (max_cells && n_blocks) {
        uint32_t* bad = ctx->arena.get<uint32_t>(2); dzero(bad, 8, ctx->stream);
        SKH_LAUNCH(screen_add_cells_kernel, dim3((unsigned)((max_cells + 255) / 256), n_blocks), 256, 0, ctx->stream, d_blocks, block_words, N, cnt, bad);
        check_launch("screen_add_cells");
        uint32_t h_bad = 0; d2h(&h_bad, bad, 4, ctx->stream);
        if (h_bad) throw std::invalid_argument("screen_from_cells: a cell names a genome beyond the set");
    }
    threshold_rows(ctx, cnt, 1, plane, 0, N, N, sr, S->d_mk_off.p, S->d_mk_off.p, first, second, nullptr);
    dsync(ctx->stream);
}

}  // namespace skh
