// chain_select.h -- get_nonoverlapping_chains (chain.rs:1008-1099): greedy_fast_kernel / greedy_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ greedy selection
__device__ __forceinline__ int ivl_cmp(const Interval& a, const Interval& b) {     // derived PartialOrd over the field order
#define SKH_CMP(f) if (a.f != b.f) return a.f < b.f ? -1 : 1;
    SKH_CMP(score) SKH_CMP(na) SKH_CMP(q0) SKH_CMP(q1) SKH_CMP(r0) SKH_CMP(r1) SKH_CMP(rctg) SKH_CMP(qctg) SKH_CMP(chunk) SKH_CMP(rev)
#undef SKH_CMP
    return 0;
}

constexpr uint32_t GREEDY_LDS = 2048;   // sorted-index slots per wave kept in LDS by the fallback kernel
constexpr uint32_t GREEDY_FAST = 1023;  // pairs with at most this many candidate intervals take the all-LDS kernel (its 10-bit links keep 1023 for "none")
constexpr uint32_t GREEDY_REDO = 0xFFFFFFFFu;   // n_accepted value by which the all-LDS kernel hands a pair over to greedy_kernel

// Fast path (n <= GREEDY_FAST candidates): ONE WAVE PER PAIR AND WORKGROUP, everything staged in LDS.
//   1. bitonic sort of (key, index) with key = score(24) | anchors(20) | top 20 bits of q0; key ties (rare) fall back to
//      the full tuple comparison -> the reference's descending order (chain.rs:1012);
//   2. greedy acceptance 64 candidates at a time: every lane owns one candidate and sums its overlaps against the
//      accepted list (uniform LDS broadcasts, no reductions); the 64 decisions are then resolved in order, an accepted
//      candidate's interval being broadcast (v_readlane) to the later lanes of the same batch (chain.rs:1017-1095).
//   The greedy loop is a chain of dependent instructions, and other waves are the only thing that can fill its issue slots: what the kernel costs is
//   set by how many pairs are resident, i.e. by LDS per pair.  Round 3: an accepted interval takes 24 bytes (was 32: lengths instead of end points in
//   16 bits each, the chunk instead of the query contig -- intervals of different chunks never overlap on the query axis --, three 10-bit links and the
//   candidate's number in one word each), the sorted order 2 bytes per candidate (was 4), one wave per workgroup: 14 KB for the 512 class instead of
//   19 KB per wave -> 2.75 instead of 2 waves per SIMD.  A pair with an interval that does not fit the 16-bit fields (a chain spanning 64 kb of the
//   reference, a 65,536th chunk) is handed to greedy_kernel (GREEDY_REDO).
//   The kernel is instantiated for CAP = 256 / 512 / 1024 and a pair runs in the smallest one that holds it.
//   Pairs are handed out by decreasing candidate count (greedy_order_keys_kernel + a 16-bit radix sort): the kernel ends when its
//   slowest wave does, so the long ones start first.
// Accepted interval, 24 B, threaded on up to three lists: the accepted intervals of its chunk (query axis) and
// those of the one or two GREEDY_BIN-sized bins of the reference axis it touches (intervals spanning more go on a separate short list)
struct AccIvl { uint32_t rctg, r0, q0, lens /* r1 - r0 | (q1 - q0) << 16 */, cc /* chunk | candidate << 16 */, links /* qnext | rnext0 << 10 | rnext1 << 20 */; };
constexpr uint32_t GREEDY_NIL = 0x3FFu;         // "no entry" in the 10-bit links and in the list heads
constexpr uint32_t GREEDY_BIN_SHIFT = 15;       // 32 kb reference bins: a chain interval of a 20 kb chunk touches one or two
constexpr uint32_t GREEDY_BUCKETS = 256;        // list heads per axis (hashed chunk id / hashed (contig, bin))
constexpr uint32_t GREEDY_LONG = 64;            // accepted intervals spanning more than two bins (beyond that: every candidate scans the whole list)
__device__ __forceinline__ uint32_t greedy_rhash(uint32_t rctg, uint32_t bin) { return (rctg * 37u + bin) & (GREEDY_BUCKETS - 1u); }
// heads[bucket] <- value, returns the previous head; the 16-bit heads are exchanged through a compare-and-swap on their 32-bit word
__device__ __forceinline__ uint32_t greedy_push(uint16_t* heads, uint32_t bucket, uint32_t value) {
    unsigned* w = (unsigned*)heads + (bucket >> 1); const uint32_t sh = (bucket & 1u) * 16u;
    unsigned seen = *w, prev;
    do { prev = seen; seen = atomicCAS(w, prev, (prev & ~(0xFFFFu << sh)) | (value << sh)); } while (seen != prev);
    return (prev >> sh) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t greedy_last_bin(uint32_t r0, uint32_t r1) { const uint32_t b0 = r0 >> GREEDY_BIN_SHIFT, b1 = (r1 ? r1 - 1u : 0u) >> GREEDY_BIN_SHIFT; return b1 > b0 ? b1 : b0; }
__global__ __launch_bounds__(256) void greedy_order_keys_kernel(uint32_t n_pairs, const uint32_t* ivl_cnt, uint32_t* keys, uint32_t* vals) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t n = ivl_cnt[p];
    keys[p] = 0xFFFFu - (n > 0xFFFFu ? 0xFFFFu : n); vals[p] = p;
}
// The same order for up to GREEDY_ORDER_ONE_MAX pairs in one launch: a counting sort in LDS by min(candidates, 2047), most candidates first.  Pairs of one class come
// out in no particular order -- the order only decides which pair starts when.
constexpr uint32_t GREEDY_ORDER_ONE_MAX = 1u << 17, GREEDY_ORDER_CLASSES = 2048;
__global__ __launch_bounds__(1024) void greedy_order_kernel(uint32_t n_pairs, const uint32_t* ivl_cnt, uint32_t* order) {
    __shared__ uint32_t cls[GREEDY_ORDER_CLASSES];
    __shared__ uint32_t wsum[16];
    for (uint32_t x = threadIdx.x; x < GREEDY_ORDER_CLASSES; x += 1024) cls[x] = 0;
    __syncthreads();
    auto klass = [&](uint32_t p) { const uint32_t n = ivl_cnt[p]; return GREEDY_ORDER_CLASSES - 1u - (n < GREEDY_ORDER_CLASSES ? n : GREEDY_ORDER_CLASSES - 1u); };   // class 0 = the most candidates
    for (uint32_t p = threadIdx.x; p < n_pairs; p += 1024) atomicAdd(&cls[klass(p)], 1u);
    __syncthreads();
    // exclusive scan of the 2048 class sizes: two per thread
    const uint32_t a = cls[2 * threadIdx.x], b = cls[2 * threadIdx.x + 1];
    const uint32_t incl = wave_incl_scan(a + b), w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    if (l == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < w; i++) base += wsum[i];
    const uint32_t off = base + incl - (a + b);
    cls[2 * threadIdx.x] = off; cls[2 * threadIdx.x + 1] = off + a;
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < n_pairs; p += 1024) order[atomicAdd(&cls[klass(p)], 1u)] = p;
}
template <uint32_t CAP>
__global__ __launch_bounds__(64) void greedy_fast_kernel(uint32_t n_pairs, const uint32_t* order, const uint32_t* pi0, const uint32_t* pc0, const uint32_t* ivl_cnt,
                                                         const Interval* ivls, uint32_t len_limit, uint32_t big_min, uint32_t* ivl_next, uint32_t* chunk_head, uint32_t* n_accepted) {
    __shared__ __attribute__((aligned(16))) AccIvl acc[CAP];          // accepted intervals; the sort keys (8 B each) borrow this space first
    __shared__ uint16_t idx[CAP];                                     // the candidates in sorted order
    __shared__ __attribute__((aligned(4))) uint16_t qh[GREEDY_BUCKETS], rh[GREEDY_BUCKETS];   // pairs of heads are exchanged as 32-bit words
    __shared__ uint16_t lng[GREEDY_LONG];
    __shared__ __attribute__((aligned(4))) uint8_t cq[GREEDY_BUCKETS], cr[GREEDY_BUCKETS];   // candidates of the current batch per (hashed) chunk / reference bin; zero between batches
    if (blockIdx.x >= n_pairs) return;
    const uint32_t p = order[blockIdx.x];
    const uint32_t l = lane_id();
    const uint32_t I0 = pi0[p];
    uint32_t n = ivl_cnt[p]; const uint32_t cap = pi0[p + 1] - I0; if (n > cap) n = cap;
    if (n > GREEDY_FAST || n > CAP || (CAP > 256 && n <= CAP / 2) || n >= big_min) return;   // another instantiation's (or greedy_kernel's / greedy_big_kernel's) pair
    if (n == 0) { if (l == 0) n_accepted[p] = 0; return; }
    uint32_t N = 1; while (N < n) N <<= 1;
    unsigned long long* key = (unsigned long long*)acc;
    const Interval* iv = ivls + I0;
    bool wide_field = false;                                                        // an interval that does not fit the packed record
    for (uint32_t i = l; i < N; i += 64) {
        unsigned long long kx = 0; uint32_t ix = 0xFFFFu;
        if (i < n) {
            const Interval e = iv[i]; kx = ((unsigned long long)e.score << 40) | ((unsigned long long)(e.na & 0xFFFFFu) << 20) | (e.q0 >> 12); ix = i;
            wide_field = wide_field || e.r1 - e.r0 >= len_limit || e.q1 - e.q0 >= len_limit || e.chunk >= 0x10000u;
        }
        key[i] = kx; idx[i] = (uint16_t)ix;
    }
    if (__any(wide_field)) { if (l == 0) n_accepted[p] = GREEDY_REDO; return; }
    wave_sync_mem();
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = l; t < N / 2; t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;   // the t-th compare-exchange pair of this pass
                const uint32_t a = idx[i], b = idx[x];
                const unsigned long long ka = key[i], kb = key[x];
                const bool up = (i & k) == 0;
                // first/second: swap iff `first` must precede `second` in the final (descending, padding last) order
                const uint32_t f = up ? b : a, s2 = up ? a : b;
                const unsigned long long kf = up ? kb : ka, ks = up ? ka : kb;
                bool sw;
                if (f == 0xFFFFu) sw = false; else if (s2 == 0xFFFFu) sw = true;
                else if (kf != ks) sw = kf > ks; else sw = ivl_cmp(iv[f], iv[s2]) > 0;
                if (sw) { idx[i] = (uint16_t)b; idx[x] = (uint16_t)a; key[i] = kb; key[x] = ka; }
            }
            wave_sync_mem();
        }
    }
    for (uint32_t i = l; i < GREEDY_BUCKETS; i += 64) { qh[i] = (uint16_t)GREEDY_NIL; rh[i] = (uint16_t)GREEDY_NIL; cq[i] = 0; cr[i] = 0; }
    wave_sync_mem();
    uint32_t nacc = 0, nlong = 0;
    bool long_overflow = false;                                                     // more than GREEDY_LONG wide intervals: fall back to scanning everything
    // the next batch's candidate records are fetched while the current batch is decided (a round trip to memory per batch otherwise)
    uint32_t ci_next = l < n ? idx[l] : 0;
    Interval c_next = iv[ci_next];
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t s = base + l;
        const uint32_t ci = ci_next;
        const Interval c = c_next;
        if (base + 64 < n) { ci_next = s + 64 < n ? idx[s + 64] : 0; c_next = iv[ci_next]; }
        uint32_t sum_r = 0, sum_q = 0, cnt_r = 0, cnt_q = 0;
        auto add_r = [&](const AccIvl& a) {                                        // chain.rs:1030-1045
            const uint32_t a_r1 = a.r0 + (a.lens & 0xFFFFu);
            const bool hr = a.rctg == c.rctg && a.r0 < c.r1 && c.r0 < a_r1;
            const uint32_t xr = c.r1 - a.r0, yr = a_r1 - c.r0;
            cnt_r += hr ? 1u : 0u; sum_r += hr ? (xr < yr ? xr : yr) : 0u;
        };
        auto add_q = [&](const AccIvl& a) {                                        // chain.rs:1059-1073; same chunk <=> the only accepted intervals that can overlap on this axis
            const uint32_t a_q1 = a.q0 + (a.lens >> 16);
            const bool hq = (a.cc & 0xFFFFu) == c.chunk && a.q0 < c.q1 && c.q0 < a_q1;
            const uint32_t xq = c.q1 - a.q0, yq = a_q1 - c.q0;
            cnt_q += hq ? 1u : 0u; sum_q += hq ? (xq < yq ? xq : yq) : 0u;
        };
        if (long_overflow) {
            for (uint32_t a = 0; a < nacc; a++) { const AccIvl e = acc[a]; add_r(e); add_q(e); }   // uniform index: LDS broadcast
        } else {
            // Accepted intervals that can overlap this candidate: on the query axis those of its own chunk (chunks are disjoint ranges of one
            // contig), on the reference axis those sharing a bin with it.  An interval listed in two bins is counted in the bin that holds
            // max(candidate start, interval start), a point of the overlap if there is one.
            for (uint32_t a = qh[c.chunk & (GREEDY_BUCKETS - 1u)]; a != GREEDY_NIL;) { const AccIvl e = acc[a]; add_q(e); a = e.links & GREEDY_NIL; }
            const uint32_t c0 = c.r0 >> GREEDY_BIN_SHIFT, c1 = greedy_last_bin(c.r0, c.r1);
            for (uint32_t x = c0; x <= c1; x++) {
                const uint32_t h = greedy_rhash(c.rctg, x);
                for (uint32_t a = rh[h]; a != GREEDY_NIL;) {
                    const AccIvl e = acc[a];
                    const uint32_t e0 = e.r0 >> GREEDY_BIN_SHIFT;
                    const bool first = greedy_rhash(e.rctg, e0) == h;               // which of the interval's (at most two, consecutive) bins hangs on this head
                    const uint32_t eb = first ? e0 : e0 + 1u;
                    if (e.rctg == c.rctg && eb == x && x == (c0 > e0 ? c0 : e0)) add_r(e);
                    a = (first ? e.links >> 10 : e.links >> 20) & GREEDY_NIL;
                }
            }
            for (uint32_t t = 0; t < nlong; t++) add_r(acc[lng[t]]);
        }
        const uint32_t nb = n - base < 64 ? n - base : 64, nacc0 = nacc;
        // Which candidates of the batch can influence each other at all?  Two intervals overlap on the query axis only inside one chunk, on the reference
        // axis only if they share a 32 kb bin: every candidate counts itself into a (hashed) counter of its chunk and of its one or two bins, and one whose
        // counters all read 1 shares neither with another candidate of the batch -- its decision is the one the accepted lists gave it, and it is taken for
        // all such candidates AT ONCE.  Only the others (~a third) go through the sequential loop below, in their order.  (An interval over three or more
        // bins is always taken sequentially; hash collisions only send more candidates there.)  The counters are taken back afterwards: no clearing pass.
        const bool in_batch = l < nb;
        const uint32_t cb0 = c.r0 >> GREEDY_BIN_SHIFT, cb1 = greedy_last_bin(c.r0, c.r1);
        const uint32_t n_bins = cb1 - cb0 + 1u < 4u ? cb1 - cb0 + 1u : 4u;           // (an interval here is shorter than greedy_len_limit = 64 kb: at most three bins)
        const uint32_t kq = c.chunk & (GREEDY_BUCKETS - 1u);
        auto bump = [&](uint8_t* t, uint32_t k, int d) { atomicAdd((unsigned*)t + (k >> 2), (unsigned)d << ((k & 3u) * 8u)); };
        if (in_batch) { bump(cq, kq, 1); for (uint32_t x = 0; x < n_bins; x++) bump(cr, greedy_rhash(c.rctg, cb0 + x), 1); }
        wave_sync_mem();
        bool alone = in_batch && n_bins <= 2u && cb1 - cb0 < 2u && cq[kq] == 1;
        if (alone) {
            const uint32_t k0 = greedy_rhash(c.rctg, cb0), k1 = greedy_rhash(c.rctg, cb1);
            alone = n_bins == 1u ? cr[k0] == 1 : (k0 == k1 ? cr[k0] == 2 : (cr[k0] == 1 && cr[k1] == 1));
        }
        wave_sync_mem();
        if (in_batch) { bump(cq, kq, -1); for (uint32_t x = 0; x < n_bins; x++) bump(cr, greedy_rhash(c.rctg, cb0 + x), -1); }
        {   // the candidates that stand alone: accepted or not by the sums they have, all at once
            const bool ok_r = cnt_r == 0 || (float)sum_r < (float)(c.r1 - c.r0) * 0.5f;   // chain.rs:1046 OVERLAP_ORTHOLOGOUS_FRACTION
            const bool ok_q = cnt_q == 0 || (float)sum_q < (float)(c.q1 - c.q0) * 0.5f;   // chain.rs:1075
            const bool take = alone && ok_r && ok_q;
            const unsigned long long tm = __ballot(take);
            if (take) {
                const uint32_t slot = nacc + (uint32_t)__popcll(tm & ((1ull << l) - 1ull));
                acc[slot] = AccIvl{c.rctg, c.r0, c.q0, (c.r1 - c.r0) | ((c.q1 - c.q0) << 16), c.chunk | (ci << 16), GREEDY_NIL | (GREEDY_NIL << 10) | (GREEDY_NIL << 20)};
            }
            nacc += (uint32_t)__popcll(tm);
        }
        for (unsigned long long rest = __ballot(in_batch && !alone); rest; rest &= rest - 1ull) {
            const uint32_t b = (uint32_t)__ffsll((long long)rest) - 1u;
            const bool ok_r = cnt_r == 0 || (float)sum_r < (float)(c.r1 - c.r0) * 0.5f;   // chain.rs:1046 OVERLAP_ORTHOLOGOUS_FRACTION
            const bool ok_q = cnt_q == 0 || (float)sum_q < (float)(c.q1 - c.q0) * 0.5f;   // chain.rs:1075
            const int okb = wave_readlane((int)((ok_r && ok_q) ? 1 : 0), (int)b);
            if (okb) {                                                             // wave-uniform
                const uint32_t actg = wave_readlane(c.rctg, (int)b), ar0 = wave_readlane(c.r0, (int)b), ar1 = wave_readlane(c.r1, (int)b);
                const uint32_t aqc = wave_readlane(c.qctg, (int)b), aq0 = wave_readlane(c.q0, (int)b), aq1 = wave_readlane(c.q1, (int)b);
                const uint32_t bci = wave_readlane(ci, (int)b), bchunk = wave_readlane(c.chunk, (int)b);
                if (l > b) {                                                       // later candidates of this batch see the new accepted interval
                    const bool hr = actg == c.rctg && ar0 < c.r1 && c.r0 < ar1;
                    const bool hq = aqc == c.qctg && aq0 < c.q1 && c.q0 < aq1;
                    const uint32_t xr = c.r1 - ar0, yr = ar1 - c.r0, xq = c.q1 - aq0, yq = aq1 - c.q0;
                    cnt_r += hr ? 1u : 0u; sum_r += hr ? (xr < yr ? xr : yr) : 0u;
                    cnt_q += hq ? 1u : 0u; sum_q += hq ? (xq < yq ? xq : yq) : 0u;
                }
                const uint32_t b0 = ar0 >> GREEDY_BIN_SHIFT, b1 = greedy_last_bin(ar0, ar1);
                const bool wide = b1 - b0 >= 2u;                                   // wave-uniform, like everything about the accepted interval
                if (l == 0) {                                                      // stores only: nothing in this loop waits for LDS
                    acc[nacc] = AccIvl{actg, ar0, aq0, (ar1 - ar0) | ((aq1 - aq0) << 16), bchunk | (bci << 16), GREEDY_NIL | (GREEDY_NIL << 10) | (GREEDY_NIL << 20)};
                    if (wide && nlong < GREEDY_LONG) lng[nlong] = (uint16_t)nacc;
                }
                if (wide) { if (nlong < GREEDY_LONG) nlong++; else long_overflow = true; }
                nacc++;
            }
        }
        wave_sync_mem();
        // link this batch's accepted intervals into the lists, one per lane (the lists' order is free)
        if (nacc0 + l < nacc) {
            AccIvl* e = &acc[nacc0 + l];
            const uint32_t r1 = e->r0 + (e->lens & 0xFFFFu);
            uint32_t links = greedy_push(qh, e->cc & (GREEDY_BUCKETS - 1u), nacc0 + l) | (GREEDY_NIL << 10) | (GREEDY_NIL << 20);
            const uint32_t b0 = e->r0 >> GREEDY_BIN_SHIFT, b1 = greedy_last_bin(e->r0, r1);
            if (b1 - b0 < 2u) {
                links = (links & ~(GREEDY_NIL << 10)) | (greedy_push(rh, greedy_rhash(e->rctg, b0), nacc0 + l) << 10);
                if (b1 > b0) links = (links & ~(GREEDY_NIL << 20)) | (greedy_push(rh, greedy_rhash(e->rctg, b1), nacc0 + l) << 20);
            }
            e->links = links;
        }
        wave_sync_mem();
    }
    // good_non_overlap_intervals[chunk_id].push (chain.rs:1086-1094) for all accepted intervals at once: the per-chunk lists are only ever
    // summed over (chunk_stats_kernel), so their order is free -- and a push from inside the loop above would put a global-memory round trip
    // (read the chunk's head) into every one of the ~400 sequential steps of a pair
    for (uint32_t a = l; a < nacc; a += 64) {
        const uint32_t cc = acc[a].cc, bci = cc >> 16;
        const uint32_t slot = pc0[p] + (cc & 0xFFFFu);
        ivl_next[I0 + bci] = atomicExch(&chunk_head[slot], I0 + bci);
    }
    if (l == 0) n_accepted[p] = nacc;
}

// Fallback for pairs with GREEDY_FAST < n <= GREEDY_LDS candidate intervals, and for those the fast kernel handed over: one wave per pair: bitonic-sort the pair's candidate
// interval indices into DESCENDING tuple order (chain.rs:1012), then accept greedily (chain.rs:1017-1095), every candidate against all accepted
// ones.  An accepted interval is flagged in bit 31 of its sorted slot and pushed on its chunk's list.  (Pairs beyond GREEDY_LDS: greedy_big_kernel.)
__global__ __launch_bounds__(256) void greedy_kernel(uint32_t n_pairs, const uint32_t* pi0, const uint32_t* pc0, const uint32_t* ivl_cnt,
                                                     const Interval* ivls, uint32_t big_min, uint32_t* ivl_next, uint32_t* chunk_head, uint32_t* n_accepted) {
    __shared__ uint32_t lds_idx[4][GREEDY_LDS];
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + wv;
    if (p >= n_pairs) return;
    const uint32_t l = lane_id();
    const uint32_t I0 = pi0[p];
    uint32_t n = ivl_cnt[p]; const uint32_t cap = pi0[p + 1] - I0; if (n > cap) n = cap;
    if (n >= big_min) return;                                                       // greedy_big_kernel's pair
    if (n <= GREEDY_FAST && n_accepted[p] != GREEDY_REDO) return;                   // done by greedy_fast_kernel (which writes n_accepted for every pair it is given)
    uint32_t N = 1; while (N < n) N <<= 1;                                          // n < big_min <= GREEDY_LDS + 1
    uint32_t* idx = lds_idx[wv];
    const Interval* iv = ivls + I0;
    for (uint32_t i = l; i < N; i += 64) idx[i] = i < n ? i : NONE;
    wave_sync_mem();
    // before(a,b): a precedes b in the final order (greater tuple first; padding last)
    for (uint32_t k = 2; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = l; i < N; i += 64) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const uint32_t a = idx[i], b = idx[x];
                    const bool up = (i & k) == 0;
                    const uint32_t first = up ? b : a, second = up ? a : b;          // swap iff `first` must precede `second`
                    const bool sw = first != NONE && (second == NONE || ivl_cmp(iv[first], iv[second]) > 0);
                    if (sw) { idx[i] = b; idx[x] = a; }
                }
            }
            wave_sync_mem();
        }
    }
    uint32_t nacc = 0;
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t ci = idx[s] & 0x7FFFFFFFu;
        const Interval c = iv[ci];
        uint32_t sum_r = 0, sum_q = 0, cnt_r = 0, cnt_q = 0;
        for (uint32_t t = l; t < s; t += 64) {
            const uint32_t e = idx[t];
            if (e & 0x80000000u) {
                const Interval o = iv[e & 0x7FFFFFFFu];
                if (o.rctg == c.rctg && o.r0 < c.r1 && c.r0 < o.r1) { cnt_r++; const uint32_t x = c.r1 - o.r0, y = o.r1 - c.r0; sum_r += x < y ? x : y; }   // chain.rs:1036-1045
                if (o.qctg == c.qctg && o.q0 < c.q1 && c.q0 < o.q1) { cnt_q++; const uint32_t x = c.q1 - o.q0, y = o.q1 - c.q0; sum_q += x < y ? x : y; }   // chain.rs:1065-1073
            }
        }
        sum_r = wave_sum(sum_r); sum_q = wave_sum(sum_q); cnt_r = wave_sum(cnt_r); cnt_q = wave_sum(cnt_q);
        const bool ok_r = cnt_r == 0 || (float)sum_r < (float)(c.r1 - c.r0) * 0.5f;      // chain.rs:1046 OVERLAP_ORTHOLOGOUS_FRACTION
        const bool ok_q = cnt_q == 0 || (float)sum_q < (float)(c.q1 - c.q0) * 0.5f;      // chain.rs:1075
        if (ok_r && ok_q) {
            if (l == 0) {
                idx[s] = ci | 0x80000000u;
                const uint32_t slot = pc0[p] + c.chunk;
                ivl_next[I0 + ci] = chunk_head[slot]; chunk_head[slot] = I0 + ci;
            }
            nacc++;
        }
        wave_sync_mem();
    }
    if (l == 0) n_accepted[p] = nacc;
}

// Pairs with GREEDY_LDS or more candidate intervals (a 2.3 Gbp genome against its relative: 130,000): one workgroup per pair, everything in a global
// scratch area of GREEDY_BIG_WORDS words per candidate.  (greedy_kernel's every-candidate-against-every-accepted-one walk took 105 s for that pair.)
//   1. all 1024 threads sort (key, candidate) records into the reference's order (chain.rs:1012) with a bitonic network whose comparators all point
//      the same way, so that any n works without padding; key = the leading fields of the tuple, inverted; equal keys compare the whole tuple;
//   2. wave 0 accepts greedily (chain.rs:1017-1095) 64 candidates at a time like greedy_fast_kernel.  The accepted intervals a candidate can overlap
//      are found through lists: on the query axis its chunk's list -- the output list itself (chunk_head / ivl_next) --, on the reference axis the
//      lists of the GREEDY_BIN-sized bins it touches (heads hashed by (contig, bin), one head per candidate slot), plus a list of the accepted
//      intervals spanning more than two bins.  Lists are written with atomics and read past the L1.
constexpr uint32_t GREEDY_BIG_WORDS = 8;        // per candidate slot: record (key, candidate: 4 words) | bin-list links (2) | a bin-list head | a long-list entry
__device__ __forceinline__ uint32_t greedy_big_bucket(uint32_t rctg, uint32_t bin, uint32_t H) { return (uint32_t)(((uint64_t)(rctg * 0x9E3779B1u) + bin) % H); }   // consecutive bins -> different buckets
__global__ __launch_bounds__(1024) void greedy_big_kernel(uint32_t n_pairs, const uint32_t* pi0, const uint32_t* ps0, const uint32_t* pc0, const uint32_t* ivl_cnt,
                                                          const Interval* ivls, uint32_t big_min, uint32_t* scratch, uint32_t* ivl_next, uint32_t* chunk_head, uint32_t* n_accepted) {
    const uint32_t p = blockIdx.x;
    if (p >= n_pairs) return;
    const uint32_t I0 = pi0[p], cap = pi0[p + 1] - I0;
    uint32_t n = ivl_cnt[p]; if (n > cap) n = cap;
    if (n < big_min) return;
    const uint32_t tid = threadIdx.x, l = tid & 63u, NT = blockDim.x;
    const Interval* iv = ivls + I0;
    uint32_t* area = scratch + (size_t)ps0[p] * GREEDY_BIG_WORDS;
    unsigned long long* rec = (unsigned long long*)area;                             // 2 per slot: inverted key, candidate
    uint32_t* rn0 = area + 4 * (size_t)cap; uint32_t* rn1 = rn0 + cap; uint32_t* rhd = rn1 + cap; uint32_t* lng = rhd + cap;
    const uint32_t H = cap;
    auto ld32 = [](const uint32_t* q) { return __atomic_load_n(q, __ATOMIC_RELAXED); };
    auto ld64 = [](const unsigned long long* q) { return __atomic_load_n(q, __ATOMIC_RELAXED); };
    for (uint32_t i = tid; i < n; i += NT) {
        const Interval e = iv[i];
        const unsigned long long k = ((unsigned long long)e.score << 40) | ((unsigned long long)(e.na & 0xFFFFFu) << 20) | (e.q0 >> 12);
        __atomic_store_n(&rec[2 * (size_t)i], ~k, __ATOMIC_RELAXED); __atomic_store_n(&rec[2 * (size_t)i + 1], (unsigned long long)i, __ATOMIC_RELAXED);
    }
    for (uint32_t i = tid; i < H; i += NT) __atomic_store_n(&rhd[i], NONE, __ATOMIC_RELAXED);
    block_fence();
    __syncthreads();
    // ---- 1. sort
    uint32_t N = 1; while (N < n) N <<= 1;
    auto exchange = [&](uint32_t i, uint32_t x) {                                    // i < x: the smaller record to i
        if (x >= n) return;
        const unsigned long long ka = ld64(&rec[2 * (size_t)i]), kb = ld64(&rec[2 * (size_t)x]);
        if (kb > ka) return;
        const unsigned long long a = ld64(&rec[2 * (size_t)i + 1]), b = ld64(&rec[2 * (size_t)x + 1]);
        if (kb == ka && ivl_cmp(iv[b], iv[a]) <= 0) return;                          // descending tuples: b goes first only when it is the greater one
        __atomic_store_n(&rec[2 * (size_t)i], kb, __ATOMIC_RELAXED); __atomic_store_n(&rec[2 * (size_t)i + 1], b, __ATOMIC_RELAXED);
        __atomic_store_n(&rec[2 * (size_t)x], ka, __ATOMIC_RELAXED); __atomic_store_n(&rec[2 * (size_t)x + 1], a, __ATOMIC_RELAXED);
    };
    for (uint32_t k = 2; k <= N; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = tid; t < N / 2; t += NT) { const uint32_t blk = t / hk, off = t % hk; exchange(blk * k + off, blk * k + (k - 1u - off)); }   // the mirrored step
        block_fence(); __syncthreads();
        for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < N / 2; t += NT) { const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)); exchange(i, i | j); }
            block_fence(); __syncthreads();
        }
    }
    if (tid >= 64) return;
    // ---- 2. greedy acceptance (wave 0)
    uint32_t nacc = 0, nlong = 0;
    const uint32_t C0 = pc0[p];
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t s = base + l; const bool have = s < n;
        const uint32_t ci = have ? (uint32_t)ld64(&rec[2 * (size_t)s + 1]) : 0u;
        const Interval c = iv[ci];
        uint32_t sum_r = 0, sum_q = 0, cnt_r = 0, cnt_q = 0;
        auto add_r = [&](const Interval& a) {                                      // chain.rs:1030-1045
            const bool hr = a.rctg == c.rctg && a.r0 < c.r1 && c.r0 < a.r1;
            const uint32_t xr = c.r1 - a.r0, yr = a.r1 - c.r0;
            cnt_r += hr ? 1u : 0u; sum_r += hr ? (xr < yr ? xr : yr) : 0u;
        };
        auto add_q = [&](const Interval& a) {                                      // chain.rs:1059-1073 (the chunk's list: same query contig)
            const bool hq = a.q0 < c.q1 && c.q0 < a.q1;
            const uint32_t xq = c.q1 - a.q0, yq = a.q1 - c.q0;
            cnt_q += hq ? 1u : 0u; sum_q += hq ? (xq < yq ? xq : yq) : 0u;
        };
        const uint32_t c0 = c.r0 >> GREEDY_BIN_SHIFT, c1 = greedy_last_bin(c.r0, c.r1);
        if (have) {
            for (uint32_t e = ld32(&chunk_head[C0 + c.chunk]); e != NONE; e = ld32(&ivl_next[e])) add_q(ivls[e]);
            for (uint32_t x = c0; x <= c1; x++) {
                const uint32_t h = greedy_big_bucket(c.rctg, x, H);
                for (uint32_t a = ld32(&rhd[h]); a != NONE;) {
                    const Interval e = iv[a];
                    const uint32_t e0 = e.r0 >> GREEDY_BIN_SHIFT;
                    const bool first = greedy_big_bucket(e.rctg, e0, H) == h;       // which of the interval's (at most two, consecutive) bins hangs on this head
                    const uint32_t eb = first ? e0 : e0 + 1u;
                    if (e.rctg == c.rctg && eb == x && x == (c0 > e0 ? c0 : e0)) add_r(e);   // counted in the bin of max(candidate start, interval start)
                    a = ld32(first ? &rn0[a] : &rn1[a]);
                }
            }
            for (uint32_t t = 0; t < nlong; t++) add_r(iv[ld32(&lng[t])]);
        }
        const uint32_t nb = n - base < 64 ? n - base : 64;
        bool mine = false;
        for (uint32_t b = 0; b < nb; b++) {
            const bool ok_r = cnt_r == 0 || (float)sum_r < (float)(c.r1 - c.r0) * 0.5f;   // chain.rs:1046 OVERLAP_ORTHOLOGOUS_FRACTION
            const bool ok_q = cnt_q == 0 || (float)sum_q < (float)(c.q1 - c.q0) * 0.5f;   // chain.rs:1075
            const int okb = wave_readlane((int)((ok_r && ok_q) ? 1 : 0), (int)b);
            if (!okb) continue;                                                     // wave-uniform
            if (l == b) mine = true;
            const uint32_t actg = wave_readlane(c.rctg, (int)b), ar0 = wave_readlane(c.r0, (int)b), ar1 = wave_readlane(c.r1, (int)b);
            const uint32_t achunk = wave_readlane(c.chunk, (int)b), aq0 = wave_readlane(c.q0, (int)b), aq1 = wave_readlane(c.q1, (int)b);
            if (l > b) {                                                            // later candidates of this batch see the new accepted interval
                const bool hr = actg == c.rctg && ar0 < c.r1 && c.r0 < ar1;
                const bool hq = achunk == c.chunk && aq0 < c.q1 && c.q0 < aq1;
                const uint32_t xr = c.r1 - ar0, yr = ar1 - c.r0, xq = c.q1 - aq0, yq = aq1 - c.q0;
                cnt_r += hr ? 1u : 0u; sum_r += hr ? (xr < yr ? xr : yr) : 0u;
                cnt_q += hq ? 1u : 0u; sum_q += hq ? (xq < yq ? xq : yq) : 0u;
            }
        }
        // the batch's accepted intervals onto their lists, one per lane (the lists' order is free)
        const bool is_long = mine && c1 - c0 >= 2u;
        const unsigned long long ml = __ballot(is_long);
        if (mine) {
            __atomic_store_n(&ivl_next[I0 + ci], atomicExch(&chunk_head[C0 + c.chunk], I0 + ci), __ATOMIC_RELAXED);   // good_non_overlap_intervals[chunk_id].push (chain.rs:1086-1094)
            if (is_long) __atomic_store_n(&lng[nlong + (uint32_t)__popcll(ml & ((1ull << l) - 1ull))], ci, __ATOMIC_RELAXED);
            else {
                __atomic_store_n(&rn0[ci], atomicExch(&rhd[greedy_big_bucket(c.rctg, c0, H)], ci), __ATOMIC_RELAXED);
                if (c1 > c0) __atomic_store_n(&rn1[ci], atomicExch(&rhd[greedy_big_bucket(c.rctg, c1, H)], ci), __ATOMIC_RELAXED);
            }
        }
        nlong += (uint32_t)__popcll(ml); nacc += (uint32_t)__popcll(__ballot(mine));
        wave_sync_mem();
    }
    if (l == 0) n_accepted[p] = nacc;
}
