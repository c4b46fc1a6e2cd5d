// sketch_build.hip -- turns raw seeding output into the device-resident Sketch tables.
//
// Replaces the per-sketch HashMap<u32,u64> + multi_position_storage of types.rs:207-320 and the marker HashSet
// (types.rs:272) with, per genome:
//   position order : p_seed / p_g, + p_rep = 1 bit per position: its seed occurs
//                    more than index_chain_band times in this genome -- the enumeration side of the join
//                    p_g = padded genome coordinate << 1 | canonical (common.h CTG_PAD): 4 bytes instead of (pos, contig|strand)
//   seed table     : open addressing over n_buckets = 2 x positions home slots, slot = hash << 32 | position (seeds that occur once: 95 %) or
//                    | a reference into the genome's list storage `ms` (count, positions ascending) -- the probe side: ONE memory request per hit.
//                    Built slice by slice in LDS (build_tables_kernel): no sort, no global scatter.  + 1 bit per home slot: occupied.
//                    hash = mix32(seed ^ salt) (common.h table_hash: a bijection, equal hash <=> equal seed) with the genome's salt, 0 unless its seeds
//                    crowded a stretch of the hash range (a slice's slack slots overflow): such a genome is indexed again under the next salt.
//   markers        : sorted unique u64
#include <algorithm>

#include "internal.h"

namespace skh {

__device__ __forceinline__ uint32_t seg_of(const uint64_t* off, uint32_t n_seg, uint64_t i) {  // largest g with off[g] <= i
    uint32_t lo = 0, hi = n_seg;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// (pos, contig << 1 | canonical) -> padded coordinate << 1 | canonical; Co = uint32_t, or uint64_t for a wide set (internal.h)
template <class Co>
__global__ __launch_bounds__(256) void pack_positions_kernel(const uint32_t* pos, const uint32_t* cc, const uint64_t* pos_off, const uint64_t* ctg_off, uint32_t ng,
                                                             uint64_t n, const Co* goff, Co* p_g) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = seg_of(pos_off, ng, i), c = cc[i];
    p_g[i] = ((goff[ctg_off[g] + g + (c >> 1)] + pos[i]) << 1) | (c & 1u);
}
template <class Co>
__global__ __launch_bounds__(256) void unpack_positions_kernel(const Co* p_g, const uint64_t* pos_off, const uint64_t* ctg_off, uint32_t ng, uint64_t p0,
                                                               uint64_t n, const Co* goff, uint32_t* pos, uint32_t* cc) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = seg_of(pos_off, ng, p0 + i); const Co v = p_g[p0 + i];
    const Co* go = goff + ctg_off[g] + g;
    const uint32_t c = ctg_of(go, (uint32_t)(ctg_off[g + 1] - ctg_off[g]), (Co)(v >> 1));
    if (pos) pos[i] = (uint32_t)((v >> 1) - go[c]);
    if (cc) cc[i] = (c << 1) | (uint32_t)(v & 1u);
}
// the 32-bit position records of a wide set's wide genomes: index within the genome << 1 | canonical (what their seed tables store and the join hands on)
__global__ __launch_bounds__(256) void index_positions_kernel(const uint64_t* p_g64, const uint64_t* pos_off, const uint32_t* wide_g, uint32_t ng, uint64_t n, uint32_t* p_g) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = seg_of(pos_off, ng, i);
    if (wide_g[g]) p_g[i] = ((uint32_t)(i - pos_off[g]) << 1) | (uint32_t)(p_g64[i] & 1u);
}

__global__ __launch_bounds__(256) void head_flags_kernel(const uint64_t* keys, uint64_t n, uint32_t* head) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t* src, const uint64_t* idx, uint32_t n, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// ---- seed tables, built per genome in LDS
// A genome's table is cut into slices of TAB_SLICE home slots (common.h); one workgroup builds one slice entirely in LDS and writes it out
// densely, so the build needs no sort, no global scatter and no partial-line stores (round 1 sorted all records of the batch by (genome, hash)
// with four device-wide radix passes and then emitted / placed the entries: 2.9 ms per 1000 genomes against 0.2 + 1.6 ms now).
//   first   slice_positions_kernel lists every slice's positions (index, hash); a slice's workgroup holds its positions in registers;
//   pass A  every position whose seed's home slot falls into the slice is inserted by linear probing with 64-bit LDS compare-and-swap:
//           slot = hash << 32 | multiplicity (the same seed again only bumps the count);
//   pass B  slots are classified: single / listed (2 .. band occurrences: count + 1 words of the genome's list storage, handed out by a
//           workgroup scan + one global atomic) / repetitive (more than band: the join drops the seed, chain.rs:694-696);
//   pass C  the positions are visited again: a single seed's position goes INTO its slot (a probe hit then needs no second memory request),
//           listed seeds append to their list, positions of repetitive seeds get their 'repetitive' bit (chain.rs:674-676);
//   pass D  the short lists are put into ascending order (anchors must come out in (ref contig, ref pos) order for one query position);
//   then the slice, its share of the bucket-occupancy bitmap and the number of distinct seeds are written out.
// Between passes A and B every run of occupied slots is put into ascending hash order (see there); the outcome -- which positions a seed has, in
// which order -- does not depend on the order the LDS atomics inserted in.
constexpr uint32_t BUILD_THREADS = 1024;
constexpr uint32_t SLOT_PENDING = 0xFFFFFFFEu;          // pass B -> pass C: single seed whose position is still to be filled in

// The positions of a genome, dealt to the slices their seeds' home slots fall into: (position index, hash) slice by slice, in the genome's stretch
// of p_slice (order within a slice: whatever the LDS atomics make it).  A slice's workgroup then reads its own ~2,000 positions instead of
// testing all ~40,000 of the genome (that test was a third of build_tables_kernel's instructions).  One workgroup per genome, two passes over the
// hashes: count per slice in LDS, scan, scatter.  Measured per 1000 genomes: 0.21 ms (0.26 ms with one load in flight per thread).  Variants that
// were slower: one pass into fixed-capacity lists (0.24 ms), hashes kept in registers + the list assembled in LDS and written out in order
// (0.27 ms: 80 KB of LDS leave one workgroup per CU) -- the kernel is bound by the latency of its phases, not by its scattered 8-byte writes.
// Instantiated for SLICE_LDS_MAX = 8192 slices per genome (8M positions: 32 KB of LDS counters) and, for sets with a larger genome, for SLICE_LDS_BIG = 32768
// (33M positions, a 4 Gbp genome at c = 125; 128 KB of LDS): launch `SLICES` takes the genomes with min_slices < slices <= max_slices; beyond the last launch's
// max_slices a genome's slices re-scan all its positions (mark_beyond; 3.9 s for a 2.3 Gbp pair before the second instantiation existed).
constexpr uint32_t SLICE_LDS_MAX = 8192, SLICE_LDS_BIG = 32768;
constexpr uint32_t SLICE_NO_LIST = 0xFFFFFFFFu;
constexpr uint32_t BUILD_SKIP = 0xFFFFFFFFu;            // queue_pos of a genome that is not part of the build
template <uint32_t SLICES>
__global__ __launch_bounds__(1024) void slice_positions_kernel(const uint32_t* __restrict__ p_seed, const uint32_t* __restrict__ salts, const uint32_t* __restrict__ queue_pos,
                                                                      const uint64_t* __restrict__ pos_off, const uint32_t* __restrict__ n_buckets,
                                                                      const uint32_t* __restrict__ slice_first, uint32_t min_slices, uint32_t max_slices, uint32_t mark_beyond,
                                                                      uint32_t* __restrict__ sl_start, uint32_t* __restrict__ sl_cnt, uint2* __restrict__ p_slice, uint32_t* __restrict__ p_rep_clear) {
    SKH_DYN_SMEM(smem);
    uint32_t* cnt = (uint32_t*)smem;                                                 // SLICES counters
    __shared__ uint32_t lds_scan[BUILD_THREADS / 64];
    const uint32_t g = blockIdx.x, tid = threadIdx.x, l = tid & 63u, w = tid >> 6;
    if (queue_pos[g] == BUILD_SKIP) return;                                          // not part of this build (a rebuild of the genomes that overflowed)
    const uint64_t pos0 = pos_off[g]; const uint32_t P = (uint32_t)(pos_off[g + 1] - pos0), NB = n_buckets[g], salt = salts[g];
    const uint32_t n_sl = (NB + TAB_SLICE - 1) / TAB_SLICE, s0 = slice_first[g];
    // the 'repetitive' bits (set by the table build behind this kernel) start at zero: every genome clears the words whose first bit is one of its positions
    // (full builds, first launch; a rebuild of single genomes leaves the bits alone: they do not depend on the salt)
    if (p_rep_clear && min_slices == 0) for (uint64_t w = (pos0 + 31) / 32 + tid; w * 32 < pos0 + P; w += BUILD_THREADS) p_rep_clear[w] = 0;
    if (n_sl <= min_slices) return;                                                  // an earlier launch's genome
    if (n_sl > max_slices) {                                                         // (max_slices <= SLICES)
        if (mark_beyond) for (uint32_t s = tid; s < n_sl; s += BUILD_THREADS) { sl_start[s0 + s] = 0; sl_cnt[s0 + s] = SLICE_NO_LIST; }
        return;
    }
    for (uint32_t s = tid; s < n_sl; s += BUILD_THREADS) cnt[s] = 0;
    __syncthreads();
    // (four independent loads in flight per thread: the loop is otherwise one memory round trip per position)
    for (uint32_t i0 = tid; i0 < P; i0 += 4 * BUILD_THREADS) {
        uint32_t hh[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) { const uint32_t i = i0 + u * BUILD_THREADS; hh[u] = i < P ? p_seed[pos0 + i] : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) if (i0 + u * BUILD_THREADS < P) atomicAdd(&cnt[seed_bucket(table_hash(hh[u], salt), NB) >> TAB_SLICE_SHIFT], 1u);
    }
    __syncthreads();
    constexpr uint32_t PER_MAX = SLICES / BUILD_THREADS;
    const uint32_t per = (n_sl + BUILD_THREADS - 1) / BUILD_THREADS;                 // consecutive slices per thread
    uint32_t loc[PER_MAX], sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < PER_MAX; u++) { const uint32_t s = tid * per + u; loc[u] = (u < per && s < n_sl) ? cnt[s] : 0u; sum += loc[u]; }
    const uint32_t incl = wave_incl_scan(sum);
    if (l == 63) lds_scan[w] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t q = 0; q < w; q++) run += lds_scan[q];
#pragma unroll
    for (uint32_t u = 0; u < PER_MAX; u++) {
        const uint32_t s = tid * per + u;
        if (u < per && s < n_sl) { sl_start[s0 + s] = run; sl_cnt[s0 + s] = loc[u]; cnt[s] = run; run += loc[u]; }   // cnt: the slice's write cursor now
    }
    __syncthreads();
    for (uint32_t i0 = tid; i0 < P; i0 += 4 * BUILD_THREADS) {
        uint32_t hh[4], oo[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) { const uint32_t i = i0 + u * BUILD_THREADS; hh[u] = i < P ? table_hash(p_seed[pos0 + i], salt) : 0u; }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) oo[u] = i0 + u * BUILD_THREADS < P ? atomicAdd(&cnt[seed_bucket(hh[u], NB) >> TAB_SLICE_SHIFT], 1u) : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) if (i0 + u * BUILD_THREADS < P) p_slice[pos0 + oo[u]] = make_uint2(i0 + u * BUILD_THREADS, hh[u]);
    }
}

constexpr uint32_t TABLE_THREADS = 256;                 // workgroup of build_tables_kernel (slice slots x threads, ms per 1000 genomes: 4096 x 1024 1.58, 4096 x 512 1.40, 2048 x 512 1.53, 2048 x 256 1.05, 1024 x 128 1.23)
constexpr uint32_t TABLE_MATCH_MAX = 2048;              // positions of a slice its workgroup holds in registers (more: the slice re-scans the genome)
// The build's workgroup list: (genome, slice) pairs dealt to eight queues by genome -- the slices of a genome run on one XCD and share its seed
// arrays through that L2 -- written by the device from two small per-genome tables (as a host array it was a 320 KB upload read over PCIe: 0.15 ms
// in front of the build).  Entry k * 8 + x = the k-th (genome, slice) of queue x; unused entries keep genome = 0xFFFFFFFF.
struct QueueLens { uint32_t len[8], longest; };         // entries of each queue, and of the longest: the list has 8 x longest entries
__global__ __launch_bounds__(256) void table_blocks_kernel(uint32_t ng, const uint32_t* __restrict__ slice_first, const uint32_t* __restrict__ queue_pos, QueueLens ql, uint2* __restrict__ blk) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && (threadIdx.x >> 5) < 8u)                                  // the unused tail of every queue (no fill of the whole list beforehand: one launch less in front of the build)
        for (uint32_t k = ql.len[threadIdx.x >> 5] + (threadIdx.x & 31u); k < ql.longest; k += 32u) blk[(size_t)k * 8 + (threadIdx.x >> 5)] = make_uint2(0xFFFFFFFFu, 0u);
    if (g >= ng) return;
    const uint32_t n_sl = slice_first[g + 1] - slice_first[g], k0 = queue_pos[g], x = g & 7u;
    if (k0 == BUILD_SKIP) return;
    for (uint32_t s = 0; s < n_sl; s++) blk[(size_t)(k0 + s) * 8 + x] = make_uint2(g, s);
}

__global__ __launch_bounds__(TABLE_THREADS) void build_tables_kernel(const uint2* __restrict__ blk, const uint32_t* __restrict__ slice_first, const uint32_t* __restrict__ sl_start,
                                                            const uint32_t* __restrict__ sl_cnt, const uint2* __restrict__ p_slice, const uint32_t* __restrict__ p_seed, const uint32_t* __restrict__ salts, const uint32_t* __restrict__ p_g,
                                                            const uint64_t* __restrict__ pos_off, const uint32_t* __restrict__ n_buckets, const uint64_t* __restrict__ tab_off,
                                                            const uint64_t* __restrict__ bmap_off, const uint64_t* __restrict__ ms_off, uint32_t band, uint32_t match_cap, uint32_t stage_cap,
                                                            uint64_t* __restrict__ tab, uint32_t* __restrict__ bmap, uint32_t* ms, uint32_t* ms_used, uint32_t* n_distinct,
                                                            uint32_t* p_rep, uint32_t* err_g) {
    SKH_DYN_SMEM(smem);
    unsigned long long* slots = (unsigned long long*)smem;                          // TAB_SLICE + TAB_SLACK
    uint32_t* lbm = (uint32_t*)(smem + (size_t)(TAB_SLICE + TAB_SLACK) * 8);          // the slice's filter words (common.h): TAB_SLICE / TAB_FILTER_HOMES
    uint32_t* mlist = lbm + TAB_SLICE / TAB_FILTER_HOMES;                                         // stage_cap words: the slice's seed lists are assembled here
    __shared__ uint32_t lds_scan[TABLE_THREADS / 64];
    __shared__ uint32_t ms_base, distinct;
    const uint2 gs = blk[blockIdx.x];
    if (gs.x == 0xFFFFFFFFu) return;
    const uint32_t g = gs.x, sl = gs.y, tid = threadIdx.x, l = tid & 63u, w = tid >> 6;
    const uint64_t pos0 = pos_off[g]; const uint32_t P = (uint32_t)(pos_off[g + 1] - pos0), NB = n_buckets[g], salt = salts[g];
    uint32_t* const err = err_g + g;                                                 // per genome: it is indexed again under another salt (build_sketch_tables_finish)
    const uint32_t h0 = sl * TAB_SLICE, nh = (NB - h0 < TAB_SLICE ? NB - h0 : TAB_SLICE), phys = nh + TAB_SLACK;   // home slots / physical slots of this slice
    const uint64_t ms0 = ms_off[g]; const uint32_t ms_cap = (uint32_t)(ms_off[g + 1] - ms0);
    for (uint32_t a = tid; a < phys; a += TABLE_THREADS) slots[a] = TAB_EMPTY;
    for (uint32_t x = tid; x < TAB_SLICE / TAB_FILTER_HOMES; x += TABLE_THREADS) lbm[x] = 0;
    if (tid == 0) distinct = 0;
    __syncthreads();
    // ---- the slice's positions: listed by slice_positions_kernel (one per thread and round, index and hash in registers); a slice with more positions
    // than the passes hold that way (sequence with very few distinct seeds), or a genome of more slices than that kernel handles, re-scans all positions
    const uint32_t bi = slice_first[g] + sl, n_match = sl_cnt[bi];
    const bool dense = n_match <= match_cap;
    const uint32_t NM = dense ? n_match : P;
    constexpr uint32_t MAX_OWN = TABLE_MATCH_MAX / TABLE_THREADS;                    // listed positions per thread in dense mode
    uint32_t own[MAX_OWN], own_h[MAX_OWN];
    {
        const uint2* mine_list = p_slice + pos0 + sl_start[bi];
#pragma unroll
        for (uint32_t u = 0; u < MAX_OWN; u++) {
            uint2 e = make_uint2(0xFFFFFFFFu, 0u);
            if (dense && tid + u * TABLE_THREADS < NM) e = mine_list[tid + u * TABLE_THREADS];
            own[u] = e.x; own_h[u] = e.y;
        }
    }
    const uint32_t n_round = dense ? MAX_OWN : (P + TABLE_THREADS - 1) / TABLE_THREADS;
    // ---- pass A
    for (uint32_t u = 0; u < n_round; u++) {
        const uint32_t i = dense ? own[u < MAX_OWN ? u : 0] : u * TABLE_THREADS + tid;
        if (i >= P) continue;
        const uint32_t h = dense ? own_h[u < MAX_OWN ? u : 0] : table_hash(p_seed[pos0 + i], salt), home = seed_bucket(h, NB);
        if (home - h0 >= nh) continue;
        uint32_t a = home - h0;
        for (;;) {
            unsigned long long cur = slots[a];
            if (cur == TAB_EMPTY) {
                cur = atomicCAS(&slots[a], (unsigned long long)TAB_EMPTY, ((unsigned long long)h << 32) | 1ull);
                if (cur == TAB_EMPTY) { atomicOr(&lbm[(home - h0) >> TAB_FILTER_SHIFT], tab_filter_bits(h)); break; }
            }
            if ((uint32_t)(cur >> 32) == h) { atomicAdd(&slots[a], 1ull); break; }
            if (++a + 1 >= phys) { atomicOr(err, 1u); break; }                      // more than TAB_SLACK entries pushed past the slice's end (its last slot stays empty: it ends every walk)
        }
    }
    __syncthreads();
    // ---- clusters into ascending hash order.  Homes ascend with the hash, and in a run of occupied slots the k-th smallest home is never beyond the
    // k-th slot, so the sorted run is still a valid linear-probing layout -- and a probe may stop at the first larger hash instead of walking to the
    // run's end (the join's count pass: 2.5 ms against 4.3 ms with clusters in insertion order).  Sorting by rank: every occupied slot finds its
    // cluster's first slot and counts the smaller entries of the cluster (reads only, clusters are short: two slots on average at load 0.5, the
    // longest of a slice around thirty), then all entries move at once.  (A thread per cluster sorting by insertion: +0.9 ms per 1000 genomes.)
    {
        constexpr uint32_t MAX_SLOTS = (TAB_SLICE + TAB_SLACK + TABLE_THREADS - 1) / TABLE_THREADS;
        unsigned long long mine[MAX_SLOTS]; uint32_t dest[MAX_SLOTS];
#pragma unroll
        for (uint32_t u = 0; u < MAX_SLOTS; u++) {
            const uint32_t a = tid + u * TABLE_THREADS;
            mine[u] = a < phys ? slots[a] : TAB_EMPTY; dest[u] = a;
            if (mine[u] == TAB_EMPTY) continue;
            uint32_t first = a, smaller = 0;
            while (first > 0) { const unsigned long long x = slots[first - 1]; if (x == TAB_EMPTY) break; smaller += x < mine[u] ? 1u : 0u; first--; }
            for (uint32_t y = a + 1; y < phys; y++) { const unsigned long long x = slots[y]; if (x == TAB_EMPTY) break; smaller += x < mine[u] ? 1u : 0u; }
            dest[u] = first + smaller;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < MAX_SLOTS; u++) if (mine[u] != TAB_EMPTY) slots[dest[u]] = mine[u];
    }
    __syncthreads();
    // ---- pass B: thread t owns slots t, t + 1024, ... ; list storage by a workgroup scan
    uint32_t need = 0, nd = 0;
    for (uint32_t a = tid; a < phys; a += TABLE_THREADS) {
        const unsigned long long v = slots[a];
        if (v == TAB_EMPTY) continue;
        const uint32_t c = (uint32_t)v; nd++;
        if (c >= 2 && c <= band) need += c + 1;
    }
    uint32_t incl = wave_incl_scan(need);
    if (l == 63) lds_scan[w] = incl;
    nd = wave_sum(nd);
    if (l == 0 && nd) atomicAdd(&distinct, nd);
    __syncthreads();
    uint32_t before = 0, tot = 0;
    for (uint32_t q = 0; q < TABLE_THREADS / 64; q++) { const uint32_t t = lds_scan[q]; if (q < w) before += t; tot += t; }
    if (tid == 0) {
        ms_base = tot ? atomicAdd(&ms_used[g], tot) : 0u;
        if (distinct) atomicAdd(&n_distinct[g], distinct);
    }
    __syncthreads();
    const uint32_t base = ms_base;
    uint32_t off = base + before + incl - need;
    const bool ms_ok = base + tot <= ms_cap && base + tot < TAB_OFF_MASK - 8;
    if (!ms_ok && tid == 0) atomicOr(err, 2u);                                       // (bit 1: the list storage's bounds -- not a matter of the salt)
    // the lists of this slice are filled and ordered in LDS and copied out whole; only a slice with more list words than that space fills them in memory
    uint32_t* stage = mlist;
    const bool staged = tot <= stage_cap;
    for (uint32_t a = tid; a < phys; a += TABLE_THREADS) {
        const unsigned long long v = slots[a];
        if (v == TAB_EMPTY) continue;
        const uint32_t c = (uint32_t)v; uint32_t x;
        if (c == 1) x = SLOT_PENDING;
        else if (c > band || !ms_ok) x = TAB_REPETITIVE;
        else {                                                                        // list head = fill cursor now, the count in the end
            x = TAB_LISTED | ((c <= 4 ? c - 1 : 0u) << TAB_OFF_BITS) | off;
            if (staged) stage[off - base] = 0; else ms[ms0 + off] = 0;
            off += c + 1;
        }
        slots[a] = (v & 0xFFFFFFFF00000000ull) | x;
    }
    block_fence();
    __syncthreads();
    // ---- pass C
    for (uint32_t u = 0; u < n_round; u++) {
        const uint32_t i = dense ? own[u < MAX_OWN ? u : 0] : u * TABLE_THREADS + tid;
        if (i >= P) continue;
        const uint32_t h = dense ? own_h[u < MAX_OWN ? u : 0] : table_hash(p_seed[pos0 + i], salt), pg = p_g[pos0 + i];
        uint32_t a = seed_bucket(h, NB) - h0;
        if (a >= nh) continue;
        unsigned long long v = slots[a];
        while ((uint32_t)(v >> 32) != h && a + 1 < phys) v = slots[++a];             // present by construction (unless the slice overflowed: err is set)
        if ((uint32_t)(v >> 32) != h || v == TAB_EMPTY) continue;                     // (an empty slot reads as hash 0xFFFFFFFF)
        const uint32_t x = (uint32_t)v;
        if (x == TAB_REPETITIVE) { const uint64_t gi = pos0 + i; atomicOr(&p_rep[gi >> 5], 1u << (gi & 31u)); }
        else if (x == SLOT_PENDING) {
            if (pg < TAB_LISTED) slots[a] = ((unsigned long long)h << 32) | pg;
            else {                                                                    // a position beyond 2^31 does not fit beside the flag bit: list of one
                const uint32_t o = atomicAdd(&ms_used[g], 2u);
                if (o + 2 <= ms_cap && o + 2 < TAB_OFF_MASK - 8) { ms[ms0 + o] = 1; ms[ms0 + o + 1] = pg; slots[a] = ((unsigned long long)h << 32) | TAB_LISTED | o; }
                else { atomicOr(err, 2u); slots[a] = ((unsigned long long)h << 32) | TAB_REPETITIVE; }
            }
        } else if (x & TAB_LISTED) {
            const uint32_t o = x & TAB_OFF_MASK;
            if (staged) { const uint32_t q = atomicAdd(&stage[o - base], 1u); stage[o - base + 1 + q] = pg; }
            else { const uint32_t q = atomicAdd(&ms[ms0 + o], 1u); ms[ms0 + o + 1 + q] = pg; }
        }
    }
    block_fence();
    __syncthreads();
    // ---- pass D (lists into ascending order: anchors of one query position must come out by reference position) + write-out
    for (uint32_t a = tid; a < phys; a += TABLE_THREADS) {
        const unsigned long long v = slots[a];
        const uint32_t x = (uint32_t)v;
        if (v != TAB_EMPTY && (x & TAB_LISTED) && x != TAB_REPETITIVE && x != SLOT_PENDING) {
            const uint32_t o = x & TAB_OFF_MASK;
            if (staged && o >= base && o - base < tot) {
                uint32_t* L = stage + (o - base);
                const uint32_t n = L[0];
                for (uint32_t u = 2; u <= n; u++) {                                  // insertion sort of L[1..n] (n <= band)
                    const uint32_t key = L[u]; uint32_t y = u;
                    while (y > 1 && L[y - 1] > key) { L[y] = L[y - 1]; y--; }
                    L[y] = key;
                }
                for (uint32_t u = 0; u <= n; u++) ms[ms0 + o + u] = L[u];
            } else if (!staged) {
                // filled with atomics and stores of other threads; lists of other slices may share the cache lines: read and write past the L1
                uint32_t* L = ms + ms0 + o;
                const uint32_t n = __atomic_load_n(&L[0], __ATOMIC_RELAXED);
                for (uint32_t u = 2; u <= n; u++) {
                    const uint32_t key = __atomic_load_n(&L[u], __ATOMIC_RELAXED); uint32_t y = u;
                    while (y > 1) { const uint32_t prev = __atomic_load_n(&L[y - 1], __ATOMIC_RELAXED); if (prev <= key) break; __atomic_store_n(&L[y], prev, __ATOMIC_RELAXED); y--; }
                    __atomic_store_n(&L[y], key, __ATOMIC_RELAXED);
                }
            }
        }
        tab[tab_off[g] + (uint64_t)sl * (TAB_SLICE + TAB_SLACK) + a] = v;
    }
    uint32_t* gbm = bmap + bmap_off[g] + sl * (TAB_SLICE / TAB_FILTER_HOMES);
    for (uint32_t x = tid; x < TAB_SLICE / TAB_FILTER_HOMES; x += TABLE_THREADS) gbm[x] = lbm[x];   // the whole slice's words (zero beyond its home slots): the filter needs no clearing beforehand
}


void unpack_positions(skh_ctx* ctx, const skh_sketch_set* ss, uint64_t p0, uint64_t n, uint32_t* pos, uint32_t* cc) {
    if (!n || (!pos && !cc)) return;
    if (ss->wide) SKH_LAUNCH(unpack_positions_kernel<uint64_t>, (unsigned)((n + 255) / 256), 256, 0, ctx->stream, (const uint64_t*)ss->p_g64.p, (const uint64_t*)ss->d_pos_off.p,
               (const uint64_t*)ss->d_ctg_off.p, ss->n_genomes, p0, n, (const uint64_t*)ss->d_goff64.p, pos, cc);
    else SKH_LAUNCH(unpack_positions_kernel<uint32_t>, (unsigned)((n + 255) / 256), 256, 0, ctx->stream, (const uint32_t*)ss->p_g.p, (const uint64_t*)ss->d_pos_off.p,
               (const uint64_t*)ss->d_ctg_off.p, ss->n_genomes, p0, n, (const uint32_t*)ss->d_goff.p, pos, cc);
    check_launch("unpack_positions");
}

// the small per-genome tables the position kernels need on the device (export / import work without the seed tables)
void upload_set_offsets(skh_ctx* ctx, skh_sketch_set* ss) {
    const uint32_t ng = ss->n_genomes;
    if (ss->d_pos_off.n == ng + 1 && ss->d_ctg_off.n == ng + 1) return;
    ss->d_pos_off.alloc(ng + 1); h2d(ss->d_pos_off.p, ss->pos_off.data(), (ng + 1) * 8, ctx->stream);
    ss->d_goff.alloc(ss->goff.size() ? ss->goff.size() : 1); h2d(ss->d_goff.p, ss->goff.data(), ss->goff.size() * 4, ctx->stream);
    if (ss->wide) {
        ss->d_goff64.alloc(ss->goff64.size() ? ss->goff64.size() : 1); h2d(ss->d_goff64.p, ss->goff64.data(), ss->goff64.size() * 8, ctx->stream);
        std::vector<uint32_t> flags(ss->wide_g.begin(), ss->wide_g.end());
        ss->d_wide_g.alloc(ng ? ng : 1); h2d(ss->d_wide_g.p, flags.data(), (size_t)ng * 4, ctx->stream);
    }
    ss->d_ctg_off.alloc(ng + 1); h2d(ss->d_ctg_off.p, ss->ctg_off.data(), (ng + 1) * 8, ctx->stream);
}

void build_sketch_tables(skh_ctx* ctx, skh_sketch_set* ss, const uint32_t* pos, const uint32_t* cc) {
    TableBuild tb = build_sketch_tables_begin(ctx, ss, pos, cc);
    build_sketch_tables_finish(ctx, ss, tb);
}

// The kernels of a table build over the genomes listed in `only` (null: all), queued on the context's stream; returns the device counters the build
// fills: per genome an error word, the number of distinct seeds and the list words used.  The tables, their geometry and the position records exist.
static uint32_t* queue_table_build(skh_ctx* ctx, skh_sketch_set* ss, const std::vector<uint32_t>* only, DevEvent* gate = nullptr) {
    const uint32_t ng = ss->n_genomes;
    const uint64_t P = ss->pos_off[ng];
    if (!ng) return nullptr;
    std::vector<uint32_t> slice_first(ng + 1, 0);                                   // (genome, slice) -> index of the slice's position list
    std::vector<uint32_t> queue_pos(ng + 1, BUILD_SKIP); uint32_t queue_len[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (genome, slice) pairs are dealt to eight queues by genome (table_blocks_kernel)
    for (uint32_t g = 0; g < ng; g++) slice_first[g + 1] = slice_first[g] + (ss->n_buckets[g] + TAB_SLICE - 1) / TAB_SLICE;
    auto enqueue = [&](uint32_t g) { queue_pos[g] = queue_len[g & 7u]; queue_len[g & 7u] += slice_first[g + 1] - slice_first[g]; };
    if (only) for (uint32_t g : *only) enqueue(g); else for (uint32_t g = 0; g < ng; g++) enqueue(g);
    size_t n_blk = 0; for (uint32_t x = 0; x < 8; x++) n_blk = std::max<size_t>(n_blk, queue_len[x]);
    n_blk *= 8;
    // the build's small tables in ONE upload (each copy in front of the first kernel is a 5 us blit plus its launch):
    // tab_off, bmap_off, ms_off (u64 x ng+1) | slice_first, queue_pos (u32 x ng+1) | n_buckets, salt (u32 x ng) | the zeroed counters read back at the end
    const size_t n_back = 3 * (size_t)ng;
    const size_t n64 = 3 * ((size_t)ng + 1), n32 = 2 * ((size_t)ng + 1) + 2 * (size_t)ng + n_back;
    std::vector<uint64_t> pack(n64 + (n32 + 1) / 2, 0);
    uint64_t* h64 = pack.data(); uint32_t* h32 = (uint32_t*)(pack.data() + n64);
    memcpy(h64, ss->tab_off.data(), ((size_t)ng + 1) * 8); memcpy(h64 + ng + 1, ss->bmap_off.data(), ((size_t)ng + 1) * 8); memcpy(h64 + 2 * ((size_t)ng + 1), ss->ms_off.data(), ((size_t)ng + 1) * 8);
    memcpy(h32, slice_first.data(), ((size_t)ng + 1) * 4); memcpy(h32 + ng + 1, queue_pos.data(), ((size_t)ng + 1) * 4); memcpy(h32 + 2 * ((size_t)ng + 1), ss->n_buckets.data(), (size_t)ng * 4);
    memcpy(h32 + 2 * ((size_t)ng + 1) + ng, ss->salt.data(), (size_t)ng * 4);
    uint64_t* d_pack = ctx->arena.get<uint64_t>(pack.size()); h2d(d_pack, pack.data(), pack.size() * 8, ctx->stream);
    uint64_t* d_to = d_pack; uint64_t* d_bo = d_pack + ng + 1; uint64_t* d_mo = d_pack + 2 * ((size_t)ng + 1);
    uint32_t* d32 = (uint32_t*)(d_pack + n64);
    uint32_t* d_sf = d32; uint32_t* d_qp = d32 + ng + 1; uint32_t* d_nb = d32 + 2 * ((size_t)ng + 1); uint32_t* d_salt = d_nb + ng; uint32_t* d_back = d_salt + ng;
    uint2* d_blk = ctx->arena.get<uint2>(n_blk ? n_blk : 1);
    QueueLens ql; for (uint32_t x = 0; x < 8; x++) ql.len[x] = queue_len[x]; ql.longest = (uint32_t)(n_blk / 8);
    // LDS per workgroup: the slice (17 KB) + its filter words + stage_cap words in which the slice's seed lists are assembled: 22 KB, seven workgroups
    // of 256 threads per CU.  A slice takes up to match_cap positions from its list (registers); slices with more re-scan the genome.
    const uint32_t match_cap = std::min<uint32_t>(ctx->tune.build_match_cap ? ctx->tune.build_match_cap : TABLE_MATCH_MAX, TABLE_MATCH_MAX);
    const uint32_t stage_cap = ctx->tune.build_match_cap ? match_cap : 1024;       // list words of a slice assembled in LDS (~150 expected: 5 % of its ~2,000 positions are listed)
    uint32_t* d_ss = ctx->arena.get<uint32_t>(slice_first[ng] + 1); uint32_t* d_sc = ctx->arena.get<uint32_t>(slice_first[ng] + 1);
    uint2* d_ps = ctx->arena.get<uint2>(P + 1);
    // (kernels after the copies: a host-to-device copy queued behind a kernel took 130 us in the rocpd timeline of a bench step, 5 us behind another copy)
    SKH_LAUNCH(table_blocks_kernel, (ng + 255) / 256, 256, 0, ctx->stream, ng, (const uint32_t*)d_sf, (const uint32_t*)d_qp, ql, d_blk);
    check_launch("table_blocks");
    {
        const uint32_t max_a = ctx->tune.build_slice_max ? std::min<uint32_t>(ctx->tune.build_slice_max, SLICE_LDS_MAX) : SLICE_LDS_MAX;
        uint32_t most = 0; for (uint32_t g = 0; g < ng; g++) most = std::max(most, slice_first[g + 1] - slice_first[g]);
        const bool second = !ctx->tune.build_slice_max && most > SLICE_LDS_MAX;   // a genome beyond 8M positions: the instantiation with 128 KB of counters takes it
        SKH_LAUNCH(slice_positions_kernel<SLICE_LDS_MAX>, ng, BUILD_THREADS, SLICE_LDS_MAX * 4, ctx->stream, (const uint32_t*)ss->p_seed.p, (const uint32_t*)d_salt, (const uint32_t*)d_qp,
                   (const uint64_t*)ss->d_pos_off.p, (const uint32_t*)d_nb, (const uint32_t*)d_sf, 0u, max_a, second ? 0u : 1u, d_ss, d_sc, d_ps, only ? (uint32_t*)nullptr : ss->p_rep.p);
        if (second) {
            kernel_allow_lds(slice_positions_kernel<SLICE_LDS_BIG>, SLICE_LDS_BIG * 4);
            SKH_LAUNCH(slice_positions_kernel<SLICE_LDS_BIG>, ng, BUILD_THREADS, SLICE_LDS_BIG * 4, ctx->stream, (const uint32_t*)ss->p_seed.p, (const uint32_t*)d_salt, (const uint32_t*)d_qp,
                       (const uint64_t*)ss->d_pos_off.p, (const uint32_t*)d_nb, (const uint32_t*)d_sf, SLICE_LDS_MAX, SLICE_LDS_BIG, 1u, d_ss, d_sc, d_ps, (uint32_t*)nullptr);
        }
    }
    check_launch("slice_positions");
    if (n_blk) {
        const size_t lds = (size_t)(TAB_SLICE + TAB_SLACK) * 8 + TAB_SLICE / TAB_FILTER_HOMES * 4 + (size_t)stage_cap * 4;
        kernel_allow_lds(build_tables_kernel, lds);
        if (gate) gate->make_wait(ctx->stream);                                       // (internal.h MarkerBuild)
        SKH_LAUNCH(build_tables_kernel, (unsigned)n_blk, TABLE_THREADS, lds, ctx->stream, (const uint2*)d_blk, (const uint32_t*)d_sf, (const uint32_t*)d_ss, (const uint32_t*)d_sc,
                   (const uint2*)d_ps, (const uint32_t*)ss->p_seed.p, (const uint32_t*)d_salt, (const uint32_t*)ss->p_g.p,
                   (const uint64_t*)ss->d_pos_off.p, (const uint32_t*)d_nb, (const uint64_t*)d_to, (const uint64_t*)d_bo, (const uint64_t*)d_mo,
                   BP_CHAIN_BAND / ss->params.c, match_cap, stage_cap, ss->tab.p, ss->bmap.p, ss->ms.p, d_back + 2 * (size_t)ng, d_back + ng, ss->p_rep.p, d_back);
        check_launch("build_tables");
    }
    return d_back;
}

// queues the whole table build on the context's stream and returns without waiting; _finish reads the counts back
TableBuild build_sketch_tables_begin(skh_ctx* ctx, skh_sketch_set* ss, const uint32_t* pos, const uint32_t* cc, DevEvent* gate) {
    const uint32_t ng = ss->n_genomes;
    const uint64_t P = ss->pos_off[ng];
    StageTrace tr(ctx);
    // table geometry from the position counts alone (two home slots per POSITION, at least as many as per distinct seed): nothing has to come back
    // from the device before the tables are allocated
    ss->dist_off.assign(ng + 1, 0); ss->tab_off.assign(ng + 1, 0); ss->n_buckets.assign(ng, 0); ss->bmap_off.assign(ng + 1, 0); ss->ms_off.assign(ng + 1, 0);
    ss->salt.assign(ng, 0);
    uint64_t n_slices = 0;
    for (uint32_t g = 0; g < ng; g++) {
        const uint64_t pg = ss->pos_off[g + 1] - ss->pos_off[g];
        if (pg >= (1ull << 30)) throw Error("a genome with >= 2^30 seed positions does not fit the seed table's 32-bit slot fields");
        const uint32_t nb = (uint32_t)((std::max<uint64_t>(64, ss->compact ? pg + pg / 2 : 2 * pg) + TAB_FILTER_HOMES - 1) / TAB_FILTER_HOMES * TAB_FILTER_HOMES);   // whole filter words
        const uint32_t n_sl = (nb + TAB_SLICE - 1) / TAB_SLICE;
        ss->n_buckets[g] = nb;
        if ((n_slices += n_sl) >= 0xFFFFFFF0ull) throw Error("too many table slices in one build; split the batch");
        ss->tab_off[g + 1] = ss->tab_off[g] + nb + (uint64_t)n_sl * TAB_SLACK;
        ss->bmap_off[g + 1] = ss->bmap_off[g] + (((uint64_t)n_sl * TAB_SLICE / TAB_FILTER_HOMES) + 3) / 4 * 4;     // whole slices, whole 16-byte groups
        // list storage: a seed with 2 .. band positions takes one word more than it has positions (<= 1.5 words per position); genomes whose padded
        // coordinates pass 2^30 may need two words for a single position
        const uint64_t span = (ss->wide && ss->wide_g[g]) ? 0 : ss->goff[ss->ctg_off[g + 1] + g];   // (a wide genome's records are indices below 2^30)
        ss->ms_off[g + 1] = ss->ms_off[g] + (span >= (1ull << 30) ? 2 * pg : pg + pg / 2) + 16;
        if (ss->ms_off[g + 1] - ss->ms_off[g] >= 0x7FFFFFF0ull) throw Error("a genome's seed-list storage passes 2^31 words");
    }
    if (P >= 0xFFFFFFF0ull) throw Error("sketch set too large for one build (>= 2^32 seed positions); split the batch");
    ss->tab.alloc(ss->tab_off[ng] + 8);                                              // (slack behind the last table)
    ss->bmap.alloc(ss->bmap_off[ng] ? ss->bmap_off[ng] : 1); ss->ms.alloc(ss->ms_off[ng] ? ss->ms_off[ng] : 1);
    ss->p_rep.alloc(P / 32 + 1);
    // (everything above is host work on the position counts: when the seeding's compaction kernel is still running, this is where it is overlapped --
    // the first copy below may wait for the stream)
    upload_set_offsets(ctx, ss);
    if (pos || cc || ss->p_g.n != P) { ss->p_g.alloc(P); ss->indexed = false; }
    if (ss->wide && (pos || cc || ss->p_g64.n != P)) ss->p_g64.alloc(P);
    if (P > 0 && pos && cc) {
        if (ss->wide) SKH_LAUNCH(pack_positions_kernel<uint64_t>, (unsigned)((P + 255) / 256), 256, 0, ctx->stream, pos, cc, (const uint64_t*)ss->d_pos_off.p,
                   (const uint64_t*)ss->d_ctg_off.p, ng, P, (const uint64_t*)ss->d_goff64.p, ss->p_g64.p);
        SKH_LAUNCH(pack_positions_kernel<uint32_t>, (unsigned)((P + 255) / 256), 256, 0, ctx->stream, pos, cc, (const uint64_t*)ss->d_pos_off.p,
                   (const uint64_t*)ss->d_ctg_off.p, ng, P, (const uint32_t*)ss->d_goff.p, ss->p_g.p);
        check_launch("pack_positions");
    }
    if (ss->wide && P > 0 && !ss->indexed) {
        SKH_LAUNCH(index_positions_kernel, (unsigned)((P + 255) / 256), 256, 0, ctx->stream, (const uint64_t*)ss->p_g64.p, (const uint64_t*)ss->d_pos_off.p,
                   (const uint32_t*)ss->d_wide_g.p, ng, P, ss->p_g.p);
        check_launch("index_positions");
    }
    ss->indexed = true;
    TableBuild tb; tb.n = 3 * (size_t)ng;                                            // per genome: error bits, distinct seeds, list words used
    tb.d_back = queue_table_build(ctx, ss, nullptr, gate);
    tr.mark("build: seed tables queued");
    return tb;
}

// Reads the build's counters back (synchronises).  A genome whose seeds crowded one stretch of the hash range (more than TAB_SLACK entries pushed past the
// end of a slice) is indexed again under the next salt -- the reference's HashMap takes any key set (types.rs:281-320) -- by a build over those genomes only.
// returns true when what was derived from the set while the build ran is stale: a genome took another salt, or the list storage moved (chain.hip prepare_halves)
bool build_sketch_tables_finish(skh_ctx* ctx, skh_sketch_set* ss, TableBuild& tb) {
    const uint32_t ng = ss->n_genomes;
    bool moved = false;
    std::vector<uint32_t> back(tb.n, 0), distinct(ng, 0), ms_used(ng, 0), again;
    if (tb.d_back) d2h(back.data(), tb.d_back, tb.n * 4, ctx->stream);               // the build's one read-back (synchronises)
    else dsync(ctx->stream);
    for (uint32_t attempt = 0;; attempt++) {
        again.clear();
        if (attempt == 0 && ctx->tune.build_resalt_all) for (uint32_t g = 0; g < ng; g++) back[g] |= 1u;   // tests: every genome is indexed again under salt 1
        for (uint32_t g = 0; g < ng; g++) {
            if (back[g] & 2u) throw Error("seed table: a genome's seed lists do not fit their storage");
            if (back[g] & 1u) again.push_back(g); else if (attempt == 0 || ss->salt[g] == attempt) { distinct[g] = back[ng + g]; ms_used[g] = back[2 * (size_t)ng + g]; }
        }
        if (again.empty()) break;
        if (attempt >= 15) throw Error("seed table overflow: a genome's seeds crowd one stretch of the hash range under every salt tried");
        for (uint32_t g : again) ss->salt[g] = attempt + 1;
        moved = true;
        uint32_t* d_back = queue_table_build(ctx, ss, &again);
        std::fill(back.begin(), back.end(), 0u);
        d2h(back.data(), d_back, tb.n * 4, ctx->stream);
    }
    for (uint32_t g = 0; g < ng; g++) ss->dist_off[g + 1] = ss->dist_off[g] + distinct[g];
    if (ss->compact && ng) {                                                         // a resident set: the list storage shrinks to what the lists take (offsets in the slots are genome-relative)
        std::vector<uint64_t> off(ng + 1, 0), seg;
        for (uint32_t g = 0; g < ng; g++) { off[g + 1] = off[g] + ms_used[g] + 4; if (ms_used[g]) seg.insert(seg.end(), {ss->ms_off[g], off[g], (uint64_t)ms_used[g]}); }   // (+4: the join may read two words past a short list)
        DBuf<uint32_t> small(off[ng] ? off[ng] : 1);
        dzero(small.p, small.bytes(), ctx->stream);
        copy_segments(ctx, ss->ms.p, small.p, seg);
        dsync(ctx->stream);
        ss->ms = std::move(small); ss->ms_off = off;
        moved = true;
    }
    ss->tables_built = true;
    return moved;
}

void ensure_tables(skh_ctx* ctx, const skh_sketch_set* ss_c) {
    if (ss_c->tables_built) return;
    skh_sketch_set* ss = const_cast<skh_sketch_set*>(ss_c);
    std::lock_guard<std::mutex> lk(ss->build_mu);
    if (ss->tables_built) return;
    build_sketch_tables(ctx, ss, nullptr, nullptr);
}

// ---- markers: sort by (genome, marker), drop duplicates (marker_seeds is a set: seeding.rs:318, types.rs:272)
__global__ __launch_bounds__(256) void marker_keys_kernel(uint64_t* raw, const uint64_t* raw_off, uint32_t ng, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    raw[i] |= (uint64_t)seg_of(raw_off, ng, i) << 42;
}
__global__ __launch_bounds__(256) void marker_compact_kernel(const uint64_t* keys, const uint32_t* head, const uint32_t* excl, uint64_t n, uint64_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    out[excl[i]] = keys[i] & ((1ull << 42) - 1);
}

// A genome's marker set in one workgroup: its raw markers (~5,000 for 5 Mbp at m = 1000, duplicates included) are sorted in LDS by a bitonic network,
// duplicates dropped, and the set written back to the front of the genome's stretch of `raw`; uniq[g] = its size.  Replaces genome-tagged keys +
// a segmented radix sort + head flags + a scan over all markers of the batch (rocPRIM's segmented sort uses scratch memory: its launch held every
// other queue's dispatches back for ~130 us, in front of the table build -- rocpd timeline of a bench step).
constexpr uint32_t MARKER_LDS_MAX = 8192;              // raw markers of one genome the kernel sorts (more in any genome of the batch: the device-wide path)
constexpr uint32_t MARKER_THREADS = 1024;
__global__ __launch_bounds__(MARKER_THREADS) void marker_set_kernel(uint64_t* __restrict__ raw, const uint64_t* __restrict__ raw_off, uint32_t* __restrict__ uniq) {
    SKH_DYN_SMEM(smem);
    unsigned long long* key = (unsigned long long*)smem;                            // N entries, N = the power of two >= the genome's raw markers
    __shared__ uint32_t lds_scan[MARKER_THREADS / 64];
    const uint32_t g = blockIdx.x, tid = threadIdx.x, l = tid & 63u, w = tid >> 6;
    const uint64_t r0 = raw_off[g]; const uint32_t n = (uint32_t)(raw_off[g + 1] - r0);
    if (n == 0) { if (tid == 0) uniq[g] = 0; return; }
    uint32_t N = 64; while (N < n) N <<= 1;
    for (uint32_t i = tid; i < N; i += MARKER_THREADS) key[i] = i < n ? raw[r0 + i] : ~0ull;   // padding sorts last
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < N / 2; t += MARKER_THREADS) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;   // the t-th compare-exchange pair of this pass
                const unsigned long long a = key[i], b = key[x];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { key[i] = b; key[x] = a; }
            }
            __syncthreads();
        }
    }
    // distinct keys, in order: flags, workgroup scan, write-out
    constexpr uint32_t PER = MARKER_LDS_MAX / MARKER_THREADS;
    const uint32_t per = N / MARKER_THREADS ? N / MARKER_THREADS : 1;               // consecutive entries per thread (N >= 64: threads beyond N idle)
    uint32_t cnt = 0; unsigned long long mine[PER]; bool head[PER];
#pragma unroll
    for (uint32_t u = 0; u < PER; u++) {
        const uint32_t i = tid * per + u;
        head[u] = false; mine[u] = 0;
        if (u < per && i < n) { mine[u] = key[i]; head[u] = i == 0 || key[i - 1] != mine[u]; cnt += head[u] ? 1u : 0u; }
    }
    const uint32_t incl = wave_incl_scan(cnt);
    if (l == 63) lds_scan[w] = incl;
    __syncthreads();
    uint32_t at = incl - cnt, tot = 0;
    for (uint32_t q = 0; q < MARKER_THREADS / 64; q++) { const uint32_t t = lds_scan[q]; if (q < w) at += t; tot += t; }
#pragma unroll
    for (uint32_t u = 0; u < PER; u++) if (head[u]) raw[r0 + at++] = mine[u];
    if (tid == 0) uniq[g] = tot;
}
// the sets, one behind the other: markers[mk_off[g] + x] = raw[raw_off[g] + x].  One workgroup per genome: no search for the genome of an entry.
// (Rounds 2-4 also wrote the screen's incidence keys here, for a radix sort; the screen's keys are now made where they are bucketed, screen_keys.hip.)
__global__ __launch_bounds__(256) void marker_gather_kernel(const uint64_t* __restrict__ raw, const uint64_t* __restrict__ raw_off, const uint64_t* __restrict__ mk_off,
                                                            uint64_t* __restrict__ out) {
    const uint32_t g = blockIdx.x;
    const uint64_t r0 = raw_off[g], m0 = mk_off[g]; const uint32_t n = (uint32_t)(mk_off[g + 1] - m0);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out[m0 + i] = raw[r0 + i];
}

static bool marker_sets_in_lds(const skh_ctx* ctx, uint32_t ng, const std::vector<uint64_t>& raw_off, uint64_t* max_raw_out) {
    uint64_t max_raw = 0; for (uint32_t g = 0; g < ng; g++) max_raw = std::max(max_raw, raw_off[g + 1] - raw_off[g]);
    const uint32_t lds_max = ctx->tune.marker_lds_max ? std::min<uint32_t>(ctx->tune.marker_lds_max, MARKER_LDS_MAX) : MARKER_LDS_MAX;
    *max_raw_out = max_raw;
    return raw_off[ng] > 0 && max_raw <= lds_max;
}
static void launch_marker_sets(skh_ctx* ctx, uint32_t ng, DBuf<uint64_t>& raw, const std::vector<uint64_t>& raw_off, uint64_t max_raw, uint64_t** d_ro, uint32_t** d_uq) {
    *d_ro = ctx->arena.get<uint64_t>(ng + 1); h2d(*d_ro, raw_off.data(), (ng + 1) * 8, ctx->stream);
    *d_uq = ctx->arena.get<uint32_t>(ng + 1);                                         // (+ the largest bucket of the screen's incidence sort)
    uint32_t N = 64; while (N < max_raw) N <<= 1;
    const size_t lds = (size_t)N * 8;
    kernel_allow_lds(marker_set_kernel, lds);
    SKH_LAUNCH(marker_set_kernel, ng, MARKER_THREADS, lds, ctx->stream, raw.p, (const uint64_t*)*d_ro, *d_uq);
    check_launch("marker_set");
}
void build_markers_begin(skh_ctx* ctx, skh_sketch_set* ss, DBuf<uint64_t>& raw, const std::vector<uint64_t>& raw_off, MarkerBuild& mb) {
    const uint32_t ng = ss->n_genomes; uint64_t max_raw = 0;
    if (ng >= (1u << 22) || raw_off[ng] >= 0xFFFFFFF0ull || !marker_sets_in_lds(ctx, ng, raw_off, &max_raw)) return;   // (build_markers says what is wrong, or takes the device-wide passes)
    launch_marker_sets(ctx, ng, raw, raw_off, max_raw, &mb.d_ro, &mb.d_uq);
    mb.done.record(ctx->stream);
    mb.launched = true;
}
// the marker sets of a batch, on the context's stream; begun: their kernel is already queued (build_markers_begin)
void build_markers(skh_ctx* ctx, skh_sketch_set* ss, DBuf<uint64_t>& raw, const std::vector<uint64_t>& raw_off, ScreenKeysPlan* plan, uint32_t* plan_max, MarkerBuild* begun) {
    if (plan) *plan = ScreenKeysPlan{};
    const uint32_t ng = ss->n_genomes;
    const uint64_t M = raw_off[ng];
    StageTrace tr(ctx);
    ss->mk_off.assign(ng + 1, 0);
    if (ng >= (1u << 22)) throw Error("more than 4M genomes in one sketch set");
    if (M >= 0xFFFFFFF0ull) throw Error("too many markers for one build; split the batch");
    uint64_t max_raw = 0;
    if (marker_sets_in_lds(ctx, ng, raw_off, &max_raw)) {                            // one workgroup per genome, everything in LDS
        uint64_t* d_ro = nullptr; uint32_t* d_uq = nullptr;
        if (begun && begun->launched) { d_ro = begun->d_ro; d_uq = begun->d_uq; }
        else launch_marker_sets(ctx, ng, raw, raw_off, max_raw, &d_ro, &d_uq);
        // the screen's incidence sort (screen_keys.hip) counts its buckets HERE, over the sets where they were deduplicated (duplicates counted in its plan: a few per cent):
        // its largest bucket -- what decides how its second half runs -- comes back with the set sizes, and its two kernels run while the host adds those up
        if (plan && ng <= SCREEN_ID_MASK && sorted_screen_keys_fits(M)) {
            if (!ss->screen_sort) ss->screen_sort.reset(new PendingSort());
            screen_keys_count(ctx, ScreenKeysIn{raw.p, nullptr, d_ro, d_uq, ng, 0u}, M, 0, 0, ss->screen_sort.get(), d_uq + ng, *plan);
        } else dzero(d_uq + ng, 4, ctx->stream);
        std::vector<uint32_t> h_uq(ng + 1);
        d2h(h_uq.data(), d_uq, (size_t)(ng + 1) * 4, ctx->stream);
        if (plan_max) *plan_max = h_uq[ng];
        for (uint32_t g = 0; g < ng; g++) ss->mk_off[g + 1] = ss->mk_off[g] + h_uq[g];
        const uint64_t MU = ss->mk_off[ng];
        ss->markers.alloc(MU);
        ss->d_mk_off.alloc(ng + 1); h2d(ss->d_mk_off.p, ss->mk_off.data(), (ng + 1) * 8, ctx->stream);
        if (MU) {
            SKH_LAUNCH(marker_gather_kernel, ng, 256, 0, ctx->stream, (const uint64_t*)raw.p, (const uint64_t*)d_ro, (const uint64_t*)ss->d_mk_off.p, ss->markers.p);
            check_launch("marker_gather");
        }
        dsync(ctx->stream);
        tr.mark("build: markers");
        return;
    }
    if (M > 0) {                                                                     // a genome with more raw markers than the LDS sort takes: device-wide passes
        uint64_t* d_ro = ctx->arena.get<uint64_t>(ng + 1); h2d(d_ro, raw_off.data(), (ng + 1) * 8, ctx->stream);
        const unsigned nb = (unsigned)((M + 255) / 256);
        SKH_LAUNCH(marker_keys_kernel, nb, 256, 0, ctx->stream, raw.p, (const uint64_t*)d_ro, ng, M);
        check_launch("marker_keys");
        const uint64_t* sorted = sort_segments_u64(ctx, raw.p, M, ng, (const uint64_t*)d_ro, raw_off.data(), 42);   // (genome, marker) order: the raw markers are already grouped by genome
        uint32_t* head = ctx->arena.get<uint32_t>(M); uint32_t* excl = ctx->arena.get<uint32_t>(M + 1);
        SKH_LAUNCH(head_flags_kernel, nb, 256, 0, ctx->stream, sorted, M, head);
        check_launch("head_flags");
        exclusive_scan_u32(ctx, head, M, excl);
        uint32_t* d_mo = ctx->arena.get<uint32_t>(ng + 1);
        SKH_LAUNCH(gather_u32_kernel, (ng + 1 + 255) / 256, 256, 0, ctx->stream, (const uint32_t*)excl, (const uint64_t*)d_ro, ng + 1, d_mo);
        check_launch("gather_u32");
        std::vector<uint32_t> h_mo(ng + 1);
        d2h(h_mo.data(), d_mo, (ng + 1) * 4, ctx->stream);
        for (uint32_t g = 0; g <= ng; g++) ss->mk_off[g] = h_mo[g];
        ss->markers.alloc(ss->mk_off[ng]);
        SKH_LAUNCH(marker_compact_kernel, nb, 256, 0, ctx->stream, sorted, (const uint32_t*)head, (const uint32_t*)excl, M, ss->markers.p);
        check_launch("marker_compact");
    } else ss->markers.alloc(0);
    ss->d_mk_off.alloc(ng + 1); h2d(ss->d_mk_off.p, ss->mk_off.data(), (ng + 1) * 8, ctx->stream);
    dsync(ctx->stream);
    tr.mark("build: markers");
}

// host-only: per-genome contig statistics used by switch_qr (chain.rs:625-631) and the regression features
// (chain.rs:519-526): sorted contig lengths at indices n*10/100, n*50/100, n*90/100; and the padded contig starts.
void finalize_metadata(skh_sketch_set* ss) {
    const uint32_t ng = ss->n_genomes;
    // padded contig starts; a genome of wide_span padded bases or more (total length + 8192 per contig) makes the set wide
    const uint64_t lim = ss->ctx ? ss->ctx->tune.wide_span : (1ull << 31) - CTG_PAD;
    ss->goff64.assign(ss->ctg_len.size() + ng, 0);
    ss->wide = false; ss->wide_g.assign(ng, 0);
    for (uint32_t g = 0; g < ng; g++) {
        uint64_t at = CTG_PAD; const uint64_t base = ss->ctg_off[g] + g;
        for (uint64_t c = ss->ctg_off[g]; c < ss->ctg_off[g + 1]; c++) { ss->goff64[base + (c - ss->ctg_off[g])] = at; at += (uint64_t)ss->ctg_len[c] + CTG_PAD; }
        ss->goff64[base + (ss->ctg_off[g + 1] - ss->ctg_off[g])] = at;
        if (at >= lim) { ss->wide = true; ss->wide_g[g] = 1; }
    }
    ss->goff.assign(ss->goff64.size(), 0);                                            // (low words: exact for every genome that is not wide)
    for (size_t x = 0; x < ss->goff64.size(); x++) ss->goff[x] = (uint32_t)ss->goff64[x];
    if (!ss->wide) { ss->goff64.clear(); ss->wide_g.clear(); }
    ss->mean_ctg.assign(ng, 0.); ss->q10.assign(ng, 0.f); ss->q50.assign(ng, 0.f); ss->q90.assign(ng, 0.f);
    std::vector<uint32_t> v;                                                          // (one scratch vector: a heap allocation per genome was a tenth of a millisecond per 1000 genomes, on the step's critical path)
    for (uint32_t g = 0; g < ng; g++) {
        v.assign(ss->ctg_len.begin() + ss->ctg_off[g], ss->ctg_len.begin() + ss->ctg_off[g + 1]);
        if (v.empty()) continue;
        double s = 0; for (auto x : v) s += (double)x;
        ss->mean_ctg[g] = s / (double)v.size();
        if (v.size() > 1) std::sort(v.begin(), v.end());
        size_t n = v.size();
        ss->q10[g] = (float)v[n * 10 / 100]; ss->q50[g] = (float)v[n * 50 / 100]; ss->q90[g] = (float)v[n * 90 / 100];
    }
}

}  // namespace skh
