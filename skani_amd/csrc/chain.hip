// chain.hip -- chain_seeds (chain.rs:144-171) for a batch of genome pairs on gfx950.
//
// Reference stages and their GPU formulation (all integer until the final pow/mean; bit-exact stage outputs):
//   get_anchors join      chain.rs:608-737  -> join_count_kernel / join_fill_kernel
//        The enumerated sketch ("chain-query", A) is walked in POSITION order and every position probes the other
//        sketch's (B) seed index.  Because A is in (contig,pos) order and B's hash-order array is in
//        (hash(seed),contig,pos) order, anchors are PRODUCED in the reference's sorted order
//        (query_contig, query_pos, ref_contig, ref_pos, reverse) -- the reference's sort (chain.rs:721) disappears,
//        and query_positions_all (chain.rs:682-700,722-724) is just the filtered position list.
//   chunking              chain.rs:738-836  -> chunk_kernel: one wave per pair streams the anchors' query coordinates
//   chain_anchors_ani     chain.rs:838-896  -> chain_dp_thread_kernel (band <= 40): one lane per chunk, fused with
//   get_chain_intervals   chain.rs:939-1007    the interval emission; chain_dp_kernel + interval_emit_kernel (wider bands):
//                                              one wave per chunk, per-component argmax by 64-bit atomicMax
//   get_nonoverlapping    chain.rs:1008-1099-> greedy_fast_kernel / greedy_kernel: one wave per pair, bitonic sort + greedy
//   calculate_ani         chain.rs:173-555  -> chunk_stats_kernel (per chunk) + finalize_kernel (per pair, incl. CI and GBDT)
// Coordinates are padded genome coordinates (common.h CTG_PAD) throughout; contig ids reappear in the interval records.
#include <algorithm>
#include <cmath>

#include "internal.h"

namespace skh {

// ------------------------------------------------------------------------------------------------ views & descriptors
// One record per genome pair.  It carries direct pointers to the two sketches' arrays (already advanced to the genome's first
// element), so the pairs of one call may draw their sketches from any number of resident sketch sets (a sharded database).
struct PairDesc {
    // A = enumerated sketch (position order); p_g = padded coordinate << 1 | canonical
    const uint32_t *a_seed, *a_g; const uint16_t* a_cnt;
    // B = probed sketch: hash-order positions, seed index (entries, bucket directory, bucket-occupancy bitmap)
    const uint32_t* b_sg; const uint64_t* b_ent; const uint32_t *b_dir, *b_bmap;
    const uint32_t *a_goff, *b_goff;   // padded contig starts (common.h CTG_PAD), a_nctg + 1 / b_nctg + 1 entries
    uint32_t a_n;       // positions in A
    uint32_t b_nbk;     // B: buckets in its seed directory
    uint32_t flags;     // bit2: switched (chain.rs:649)
    uint32_t tile0;     // first join tile of this pair (global over the call)
    uint32_t a_nctg, b_nctg;
    // finalisation inputs (ref/query in the caller's sense, NOT A/B)
    uint32_t nctg_q, nctg_r;
    uint64_t ref_total_len, query_total_len;
    float q10_q, q50_q, q90_q, q10_r, q50_r, q90_r;
};

constexpr uint32_t JOIN_TILE = 1024;    // positions per join workgroup (256 threads x 4 rounds)
constexpr uint32_t NONE = 0xFFFFFFFFu;

// An anchor is 8 bytes in two arrays: anc_q = padded query coordinate, anc_r = padded ref coordinate << 1 | reverse_match
// (chunking only needs the first).  Contigs are recovered from the padded contig-start tables where a stage needs them
// (chunk boundaries, interval records).
struct Chunk { uint32_t a_begin, a_end, s_begin, s_end, qoff, qctg; };   // batch-relative anchor / seed-list ranges; the chunk's query contig and its padded start
struct Interval { uint32_t score, na, q0, q1, r0, r1, rctg, qctg, chunk, rev; };   // types.rs:508-519 field order = sort order

// ------------------------------------------------------------------------------------------------ join
// Workgroups are launched in "slots": slot b runs logical tile slot_tile[b] (or nothing).  The host interleaves the
// tiles so that all tiles probing the same sketch B land on the same XCD (block b -> XCD b % 8 on MI355X): B's hash
// table and seed-order arrays then stay in that XCD's 4 MiB L2 instead of being fetched by all eight.
__global__ __launch_bounds__(256) void join_count_kernel(const PairDesc* pairs, const uint32_t* slot_tile, const uint32_t* tile_pair,
                                                         uint32_t band, uint32_t* tile_anch, uint32_t* pair_anch, uint32_t* pair_inq,
                                                         uint32_t* pinfo, unsigned long long* inq_mask, uint32_t lds_words) {
    __shared__ uint32_t lds[16];
    SKH_DYN_SMEM(smem);
    uint32_t* bm = (uint32_t*)smem;
    const uint32_t tile = slot_tile[blockIdx.x];
    if (tile == NONE) return;
    const uint32_t p = tile_pair[tile];
    const PairDesc pd = pairs[p];
    const uint32_t start = (tile - pd.tile0) * JOIN_TILE;
    const uint64_t* ent = pd.b_ent; const uint32_t* dir = pd.b_dir;
    constexpr int R = JOIN_TILE / 256;
    // B's bucket-occupancy bitmap (1 bit per directory bucket, ~10 KB) is staged in LDS with coalesced 16-byte loads: 61 % of the
    // buckets are empty, and a probe of an empty bucket then costs no memory request at all.  The kernel runs at the L2's
    // request rate (one 64-byte slot per random 8-byte read), so requests are what to save.
    const uint32_t bm_words = ((pd.b_nbk + 31) / 32 + 3) / 4 * 4;
    const bool use_bm = bm_words <= lds_words;
    if (use_bm) {
        const uint4* src = (const uint4*)pd.b_bmap;
        for (uint32_t w4 = threadIdx.x; w4 < bm_words / 4; w4 += 256) ((uint4*)bm)[w4] = src[w4];
        __syncthreads();
    }
    // the four positions of this thread are probed together: their loads are independent, so they overlap
    uint32_t h[R], d0[R], d1[R]; bool live[R]; unsigned long long e[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = start + r * 256 + threadIdx.x;
        live[r] = i < pd.a_n;
        const uint32_t cnt = live[r] ? (uint32_t)pd.a_cnt[i] : 0xFFFFu;
        const uint32_t seed = live[r] ? pd.a_seed[i] : 0u;
        live[r] = live[r] && cnt <= band;                                          // chain.rs:674-676
        h[r] = mix32(seed);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        d0[r] = 0; d1[r] = 0;
        if (live[r]) {
            const uint32_t b = seed_bucket(h[r], pd.b_nbk);
            if (!use_bm || ((bm[b >> 5] >> (b & 31u)) & 1u)) { d0[r] = dir[b]; d1[r] = dir[b + 1]; }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) e[r] = d0[r] < d1[r] ? ent[d0[r]] : TAB_EMPTY;
    uint32_t na = 0, nq = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t o = r * 256 + threadIdx.x, i = start + o;
        uint32_t n_anch = 0, inq = 0, bstart = 0;
        if (live[r]) {
            unsigned long long x = e[r]; uint32_t dd = d0[r];
            // entries of a bucket ascend by hash; TAB_EMPTY (all ones) also ends the walk
            while ((uint32_t)(x >> 32) < h[r]) { dd++; x = dd < d1[r] ? ent[dd] : TAB_EMPTY; }
            if (x == TAB_EMPTY || (uint32_t)(x >> 32) != h[r]) inq = 1;            // absent in B: chain.rs:682-685
            else {
                const uint32_t cnt = (uint32_t)x & 0xFFu;
                if (cnt <= band) { inq = 1; n_anch = cnt; bstart = ((uint32_t)x >> 8) & 0xFFFFFFu; }   // else chain.rs:694-696: dropped entirely
            }
        }
        // probe record: first hit in B's hash-order array << 8 | hits (<= band <= 250); and one bit per position: "listed in
        // query_positions_all" (chain.rs:682-700), 64 positions per word straight from the ballot
        if (i < pd.a_n) pinfo[(uint64_t)tile * JOIN_TILE + o] = (bstart << 8) | n_anch;
        const unsigned long long m = __ballot(inq != 0);
        if ((threadIdx.x & 63) == 0) inq_mask[(uint64_t)tile * (JOIN_TILE / 64) + (o >> 6)] = m;
        na += n_anch; nq += inq;
    }
    na = wave_sum(na); nq = wave_sum(nq);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { lds[w] = na; lds[8 + w] = nq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ta = 0, tq = 0;
        for (uint32_t i = 0; i < 4; i++) { ta += lds[i]; tq += lds[8 + i]; }
        tile_anch[tile] = ta;
        if (ta) atomicAdd(&pair_anch[p], ta);
        if (tq) atomicAdd(&pair_inq[p], tq);
    }
}

// Emits the anchors of one tile at the offsets given by the tile scan, from the per-position probe results recorded by
// join_count_kernel (no second probe).
__global__ __launch_bounds__(256) void join_fill_kernel(const PairDesc* pairs, const uint32_t* slot_tile, const uint32_t* tile_pair,
                                                        uint32_t tile_base, const uint32_t* toff_a, const uint32_t* pinfo, uint32_t* anc_q, uint32_t* anc_r) {
    constexpr int R = JOIN_TILE / 256;
    __shared__ uint32_t lds_a[R * 4];
    const uint32_t tile = slot_tile[blockIdx.x];
    if (tile == NONE) return;
    const uint32_t lt = tile - tile_base, p = tile_pair[tile];
    const PairDesc pd = pairs[p];
    const uint32_t start = (tile - pd.tile0) * JOIN_TILE;
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    // all loads of the tile's four rounds are issued before anything depends on them; one barrier for the offsets
    uint32_t n_anch[R], qg[R], bst[R], ia[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t o = r * 256 + threadIdx.x, i = start + o;
        uint32_t c = 0; qg[r] = 0;
        if (i < pd.a_n) { c = pinfo[(uint64_t)tile * JOIN_TILE + o]; qg[r] = pd.a_g[i]; }
        n_anch[r] = c & 0xFFu; bst[r] = c >> 8;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        ia[r] = wave_incl_scan(n_anch[r]);
        if (l == 63) lds_a[r * 4 + w] = ia[r];
    }
    __syncthreads();
    uint32_t run_a = toff_a[lt];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t ba = 0, ta = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) { const uint32_t x = lds_a[r * 4 + k]; if (k < w) ba += x; ta += x; }
        if (n_anch[r]) {
            const uint32_t* bs = pd.b_sg + bst[r];
            uint32_t oa = run_a + ba + ia[r] - n_anch[r];
            for (uint32_t k = 0; k < n_anch[r]; k++, oa++) {                         // chain.rs:703-711, already in sorted order
                const uint32_t rg = bs[k];
                anc_q[oa] = qg[r] >> 1; anc_r[oa] = (rg & ~1u) | ((rg ^ qg[r]) & 1u);
            }
        }
        run_a += ta;
    }
}

// ------------------------------------------------------------------------------------------------ chunking (chain.rs:738-836)
// The reference walks the anchors once: a chunk ends at the first later anchor that leaves the contig or lies beyond the running
// end point, and a break advances the end point by exactly one CHUNK_SIZE (chain.rs:747-790); a contig change restarts it at the
// breaking anchor.  Inside one contig the end points are therefore an arithmetic progression fixed by the contig's first anchor,
//     lim_k = min(q_first + k * CHUNK_SIZE, last coordinate of the contig),            k = 1, 2, ...
// and the chunk boundaries obey  t_0 = first anchor,  t_k = max(t_{k-1} + 1, b_k)  with b_k = first anchor beyond lim_k -- an
// independent binary search per k.  Substituting u_k = t_k - k turns the recurrence into a running maximum, u_k = max(u_{k-1}, b_k - k),
// i.e. a prefix-max scan: one wave per pair handles 64 chunk boundaries per step instead of streaming every anchor.  The seed-list
// boundary of chunk k is simply the first position beyond lim_k (chain.rs:755-780); the pair's very last chunk takes the
// positions up to its last anchor instead (chain.rs:794-824).  query_positions_all is not materialised: it is the enumerated
// sketch's own position array (coordinates ascend) filtered by the join's one-bit-per-position mask, so a chunk records a range
// of POSITION indices and chunk_stats_kernel applies the mask.
__device__ __forceinline__ uint32_t first_above(const uint32_t* a, uint32_t lo, uint32_t hi, uint32_t v) {   // first index in [lo, hi) with a[i] > v, else hi
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] > v) hi = mid; else lo = mid + 1; }
    return lo;
}
__device__ __forceinline__ uint32_t lower_bound_g(const uint32_t* a, uint32_t lo, uint32_t hi, uint32_t v) {  // first index with a[i] >= v
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
// the same two searches over a sketch's position array, whose entries are coordinate << 1 | canonical
__device__ __forceinline__ uint32_t pos_first_above(const uint32_t* g1, uint32_t lo, uint32_t hi, uint32_t v) {
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((g1[mid] >> 1) > v) hi = mid; else lo = mid + 1; }
    return lo;
}
__device__ __forceinline__ uint32_t pos_lower_bound(const uint32_t* g1, uint32_t lo, uint32_t hi, uint32_t v) {
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((g1[mid] >> 1) < v) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int32_t wave_incl_max(int32_t v) {
    const uint32_t l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(v, d, 64); if (l >= (uint32_t)d) v = t > v ? t : v; }
    return v;
}

// Two-level search: every CHUNK_SAMPLE-th key of the pair's anchor / position arrays is copied to LDS once; a search first narrows its
// range [lo, hi) to one sample interval there (LDS round trips) and only the last log2(CHUNK_SAMPLE) probes go to memory.
// UPPER: first index whose key is > v; otherwise first index whose key is >= v.  samp[t] = key(array[org + t * CHUNK_SAMPLE]), t < ns.
constexpr uint32_t CHUNK_SAMPLE = 128, CHUNK_SAMPLES = 512;     // 2 x 2 KB of LDS per wave; arrays beyond 65,536 entries are searched directly
template <bool UPPER>
__device__ __forceinline__ void narrow_by_samples(const uint32_t* samp, uint32_t ns, uint32_t org, uint32_t v, uint32_t& lo, uint32_t& hi) {
    if (lo >= hi) return;
    const uint32_t t0 = (lo - org + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE;
    uint32_t t1 = (hi - org + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE; if (t1 > ns) t1 = ns;
    uint32_t a = t0, b = t1;                                                        // first sample in [t0, t1) for which the predicate holds
    while (a < b) { const uint32_t m = (a + b) >> 1; const uint32_t x = samp[m]; if (UPPER ? x > v : x >= v) b = m; else a = m + 1; }
    if (a > t0) { const uint32_t f = org + (a - 1) * CHUNK_SAMPLE + 1; lo = f > lo ? f : lo; }     // sample a-1 fails: the answer lies beyond it
    if (a < t1) { const uint32_t t = org + a * CHUNK_SAMPLE; hi = t < hi ? t : hi; }               // sample a holds: the answer is at or before it
}

__global__ __launch_bounds__(256) void chunk_kernel(uint32_t n_pairs, const PairDesc* pairs, const uint32_t* pa0,
                                                    const uint32_t* pc0, const uint32_t* anc_q,
                                                    Chunk* chunks, uint32_t* chunk_pair, uint32_t* n_chunks, uint32_t* err) {
    __shared__ uint32_t lds_samp[4][2][CHUNK_SAMPLES];
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= n_pairs) return;
    const uint32_t l = lane_id();
    const uint32_t A0 = pa0[p], A1 = pa0[p + 1], C0 = pc0[p], C1 = pc0[p + 1];
    uint32_t nc = 0;
    if (A1 > A0) {
        const uint32_t* go = pairs[p].a_goff;
        const uint32_t nctg = pairs[p].a_nctg;
        const uint32_t* ag = pairs[p].a_g; const uint32_t Q1 = pairs[p].a_n;         // the enumerated sketch's positions
        const uint32_t q_pair_last = anc_q[A1 - 1];
        const uint32_t ns_a = (A1 - A0 + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE, ns_s = (Q1 + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE;
        const bool sampled = ns_a <= CHUNK_SAMPLES && ns_s <= CHUNK_SAMPLES;
        uint32_t* sa = lds_samp[threadIdx.x >> 6][0]; uint32_t* ss = lds_samp[threadIdx.x >> 6][1];
        if (sampled) {
            for (uint32_t t = l; t < ns_a; t += 64) sa[t] = anc_q[A0 + t * CHUNK_SAMPLE];
            for (uint32_t t = l; t < ns_s; t += 64) ss[t] = ag[t * CHUNK_SAMPLE] >> 1;
            wave_sync_mem();
        }
        uint32_t sf_lo = 0, sf_hi = Q1;
        if (sampled) narrow_by_samples<true>(ss, ns_s, 0, q_pair_last, sf_lo, sf_hi);
        const uint32_t s_final = pos_first_above(ag, sf_lo, sf_hi, q_pair_last);    // the pair's final chunk ends its seed range here (chain.rs:794-824)
        // 64 query contigs per round, one per lane: the contig's anchor range [ca, ce), its first position rc0 and its number of end points;
        // then the (contig, k) items of the round are worked off 64 at a time -- a genome in a thousand contigs costs rounds of searches by the
        // sixty-fourth of its contigs, not by the contig
        uint32_t carry_cid = NONE, carry_t = 0, carry_s = 0; int32_t carry_uu = 0;
        for (uint32_t c0 = 0; c0 < nctg; c0 += 64) {
            const uint32_t cl = c0 + l; const bool cv = cl < nctg;
            const uint32_t cstart = cv ? go[cl] : 0xFFFFFFFFu, cnext = cv ? go[cl + 1] : 0xFFFFFFFFu;
            uint32_t lo_a = A0, hi_a = cv ? A1 : A0, lo_e = A0, hi_e = cv ? A1 : A0, lo_r = 0, hi_r = cv ? Q1 : 0;
            if (sampled) {
                narrow_by_samples<false>(sa, ns_a, A0, cstart, lo_a, hi_a); narrow_by_samples<false>(sa, ns_a, A0, cnext, lo_e, hi_e);
                narrow_by_samples<false>(ss, ns_s, 0, cstart, lo_r, hi_r);
            }
            while (__ballot(lo_a < hi_a || lo_e < hi_e || lo_r < hi_r) != 0ull) {  // the three searches advance together: their round trips overlap
                const uint32_t ma = (lo_a + hi_a) >> 1, me = (lo_e + hi_e) >> 1, mr = (lo_r + hi_r) >> 1;
                const uint32_t va = lo_a < hi_a ? anc_q[ma] : 0u, ve = lo_e < hi_e ? anc_q[me] : 0u, vr = lo_r < hi_r ? ag[mr] >> 1 : 0u;
                if (lo_a < hi_a) { if (va < cstart) lo_a = ma + 1; else hi_a = ma; }
                if (lo_e < hi_e) { if (ve < cnext) lo_e = me + 1; else hi_e = me; }
                if (lo_r < hi_r) { if (vr < cstart) lo_r = mr + 1; else hi_r = mr; }
            }
            const uint32_t ca = lo_a, ce = lo_e, rc0 = lo_r;                        // running_counter = 0 within the contig starts at rc0 (chain.rs:742-744)
            const bool has = cv && ce > ca;
            const uint32_t q_first = has ? anc_q[ca] : 0u, q_last = has ? anc_q[ce - 1] : 0u;
            const uint32_t kmax = has ? (q_last - q_first) / CHUNK_SIZE + 1u : 0u;  // lim_k reaches the contig's last anchor no later than this
            const uint32_t P = wave_incl_scan(kmax), M = __shfl(P, 63, 64);
            for (uint32_t j0 = 0; j0 < M; j0 += 64) {
                const uint32_t j = j0 + l; const bool iv = j < M;
                uint32_t slo = 0, shi = 63;                                        // the lane (contig) that owns item j: first with P > j
#pragma unroll
                for (int st = 0; st < 6; st++) { const uint32_t mid = (slo + shi) >> 1; const uint32_t pm = __shfl(P, (int)mid, 64); if (pm > j) shi = mid; else slo = mid + 1; }
                const int src = (int)(iv ? slo : 63u);
                const uint32_t o_kmax = __shfl(kmax, src, 64), o_P = __shfl(P, src, 64), a_c = __shfl(ca, src, 64), e_c = __shfl(ce, src, 64), r_c = __shfl(rc0, src, 64);
                const uint32_t qf = __shfl(q_first, src, 64), cn = __shfl(cnext, src, 64), cs = __shfl(cstart, src, 64);
                const uint32_t k = j - (o_P - o_kmax) + 1u;
                const uint64_t end64 = (uint64_t)qf + (uint64_t)k * CHUNK_SIZE;
                const uint32_t lim = end64 < (uint64_t)(cn - 1) ? (uint32_t)end64 : cn - 1;   // beyond it: another contig, or past the window
                //   b  = first anchor beyond lim (searching all of the pair's later anchors gives the same answer as searching the contig,
                //        because the contig's successor already lies beyond lim);  sb = first position beyond lim = seed list boundary after chunk k
                uint32_t lo_b = a_c, hi_b = iv ? A1 : a_c, lo_s = 0, hi_s = iv ? Q1 : 0;
                if (sampled) { narrow_by_samples<true>(sa, ns_a, A0, lim, lo_b, hi_b); narrow_by_samples<true>(ss, ns_s, 0, lim, lo_s, hi_s); }
                while (__ballot(lo_b < hi_b || lo_s < hi_s) != 0ull) {
                    const uint32_t mb = (lo_b + hi_b) >> 1, ms = (lo_s + hi_s) >> 1;
                    const uint32_t vb = lo_b < hi_b ? anc_q[mb] : 0u, vs = lo_s < hi_s ? ag[ms] >> 1 : 0u;
                    if (lo_b < hi_b) { if (vb > lim) hi_b = mb; else lo_b = mb + 1; }
                    if (lo_s < hi_s) { if (vs > lim) hi_s = ms; else lo_s = ms + 1; }
                }
                const uint32_t bnd = lo_b, sb = lo_s;
                const uint32_t cid = iv ? c0 + (uint32_t)src : 0xFFFFFF00u + l;    // lanes without an item: segments of their own
                int32_t v = (int32_t)bnd - (int32_t)k;                              // u_k
                if (k == 1) v = v > (int32_t)a_c ? v : (int32_t)a_c;                // u_0 = t_0 = the contig's first anchor
                if (l == 0 && cid == carry_cid) v = v > carry_uu ? v : carry_uu;    // the contig continues from the previous batch
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {                                  // running maximum within the contig
                    const int32_t tv = __shfl_up(v, d, 64); const uint32_t tc = __shfl_up(cid, d, 64);
                    if (l >= (uint32_t)d && tc == cid) v = tv > v ? tv : v;
                }
                const uint32_t t = (uint32_t)(v + (int32_t)k);                      // t_k (may run past e: the chunk is then cut at e)
                uint32_t t_prev = __shfl_up(t, 1, 64), s_prev = __shfl_up(sb, 1, 64);
                if (l == 0) { t_prev = carry_t; s_prev = carry_s; }
                if (k == 1) { t_prev = a_c; s_prev = r_c; }
                const bool valid = iv && t_prev < e_c;                              // chunk k exists
                Chunk ck; ck.a_begin = t_prev; ck.a_end = t < e_c ? t : e_c; ck.s_begin = s_prev; ck.s_end = sb; ck.qoff = cs; ck.qctg = c0 + (uint32_t)src;
                if (valid && ck.a_end == A1) ck.s_end = s_final > s_prev ? s_final : s_prev;   // the pair's final chunk
                const unsigned long long vm = __ballot(valid);
                const uint32_t slot = C0 + nc + (uint32_t)__popcll(vm & ((1ull << l) - 1ull));
                if (valid) {
                    if (slot < C1) { chunks[slot] = ck; chunk_pair[slot] = p; }
                    else atomicAdd(err, 1u);
                }
                nc += (uint32_t)__popcll(vm);
                carry_cid = __shfl(cid, 63, 64); carry_uu = __shfl(v, 63, 64); carry_t = __shfl(t, 63, 64); carry_s = __shfl(sb, 63, 64);
            }
        }
    }
    const uint32_t used = nc < C1 - C0 ? nc : C1 - C0;
    for (uint32_t s = C0 + used + l; s < C1; s += 64) { chunks[s] = Chunk{0, 0, 0, 0, 0, 0}; chunk_pair[s] = p; }
    if (l == 0) n_chunks[p] = used;
}

// Per-component argmax record kept at the component's ROOT anchor: score (24 bits) | index of the best anchor inside its
// chunk (20 bits) | number of anchors on the chain ending there (20 bits).  Max over the packed value = max score, ties ->
// largest index (chain.rs:952-964 with the set iteration order of partitions 0.2.4); 0 = "not a root".
__device__ __forceinline__ unsigned long long best_payload(uint32_t score, uint32_t local_idx, uint32_t depth) {
    return ((unsigned long long)score << 40) | ((unsigned long long)(local_idx & 0xFFFFFu) << 20) | (depth > 0xFFFFFu ? 0xFFFFFu : depth);
}
constexpr uint32_t MAX_CHUNK_ANCHORS = 1u << 20;

// ------------------------------------------------------------------------------------------------ banded chaining DP
// chain.rs:838-896 + score_anchors :558-603.  One wave per chunk.  Lanes own anchors base..base+63; sources j are
// swept in increasing order; a source's score is final when the sweep reaches it, so it is broadcast with v_readlane.
// All values are integers (positions, 20, gap) => int32 is exact where the reference uses f64.
struct Blk { uint32_t q, r, cr; int32_t score; uint32_t root, depth; };

template <int PB>
__global__ __launch_bounds__(256) void chain_dp_kernel(uint32_t n_slots, const Chunk* chunks, uint32_t band, const uint32_t* anc_q, const uint32_t* anc_r, unsigned long long* best) {
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slot >= n_slots) return;
    const Chunk ck = chunks[slot];
    if (ck.a_end <= ck.a_begin) return;
    const int l = (int)lane_id();
    Blk prev[PB];
#pragma unroll
    for (int b = 0; b < PB; b++) prev[b] = Blk{0, 0, 0, 0, 0, 0};
    for (uint32_t base = ck.a_begin; base < ck.a_end; base += 64) {
        const uint32_t t = base + (uint32_t)l;
        const bool valid = t < ck.a_end;
        Blk cur;
        uint2 av = make_uint2(0, 0);
        if (valid) av = make_uint2(anc_q[t], anc_r[t]);
        cur.q = av.x; cur.r = av.y >> 1; cur.cr = av.y & 1u;                         // cr: strand only -- different contigs are > MAX_LIN apart
        cur.score = 0; cur.root = t; cur.depth = 1;
        uint32_t ptr = t;
        uint32_t jlo = base - ck.a_begin > band ? base - band : ck.a_begin;
        const uint32_t jhi = ck.a_end < base + 64 ? ck.a_end : base + 64;
        {   // anchors ascend in q: sources more than BP_CHAIN_BAND below this block's first target cannot link to any of its targets
            const uint32_t q_base = anc_q[base];
            uint32_t lo = jlo, hi = base;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (anc_q[mid] + BP_CHAIN_BAND < q_base) lo = mid + 1; else hi = mid; }
            jlo = lo;
        }
        for (uint32_t j = jlo; j < jhi; j++) {
            uint32_t qj, rj, crj; int32_t sj;
            if (j >= base) {
                const int ln = (int)(j - base);
                // finalise lane ln: its score/ptr can no longer change (all its predecessors were swept)
                const uint32_t pj = wave_readlane(ptr, ln);
                uint32_t rootj = j, depthj = 1;
                if (pj != j) {
                    if (pj >= base) { rootj = wave_readlane(cur.root, (int)(pj - base)); depthj = wave_readlane(cur.depth, (int)(pj - base)) + 1; }
                    else {
#pragma unroll
                        for (int b = 0; b < PB; b++) {
                            const uint32_t bb = base - 64u * (uint32_t)(b + 1);
                            if (base >= 64u * (uint32_t)(b + 1) && pj >= bb && pj < bb + 64) { rootj = wave_readlane(prev[b].root, (int)(pj - bb)); depthj = wave_readlane(prev[b].depth, (int)(pj - bb)) + 1; }
                        }
                    }
                }
                if (l == ln) { cur.root = rootj; cur.depth = depthj; }
                qj = wave_readlane(cur.q, ln); rj = wave_readlane(cur.r, ln); crj = wave_readlane(cur.cr, ln); sj = wave_readlane(cur.score, ln);
            } else {
                qj = rj = crj = 0; sj = 0;
#pragma unroll
                for (int b = 0; b < PB; b++) {
                    const uint32_t bb = base - 64u * (uint32_t)(b + 1);
                    if (base >= 64u * (uint32_t)(b + 1) && j >= bb && j < bb + 64) {
                        const int ln = (int)(j - bb);
                        qj = wave_readlane(prev[b].q, ln); rj = wave_readlane(prev[b].r, ln); crj = wave_readlane(prev[b].cr, ln); sj = wave_readlane(prev[b].score, ln);
                    }
                }
            }
            // link j -> t (score_anchors).  Candidates: same ref contig and strand, i-j <= band, 0 < dq <= 2500,
            // 0 < dr <= 5000, |dr-dq| <= 300 (chain.rs:856-863, 564-597)
            if (valid && t > j && t - j <= band && cur.cr == crj) {
                const uint32_t dq = cur.q - qj;
                const bool rev = (crj & 1u) != 0;
                const bool fwd_ok = rev ? (rj > cur.r) : (cur.r > rj);
                const uint32_t dr = rev ? rj - cur.r : cur.r - rj;
                if (dq != 0 && dq <= BP_CHAIN_BAND && fwd_ok && dr <= (uint32_t)MAX_LIN) {
                    const int32_t gap = (int32_t)dr > (int32_t)dq ? (int32_t)(dr - dq) : (int32_t)(dq - dr);
                    const int32_t s = ANCHOR_SCORE - gap + sj;
                    // reference scans j downwards and replaces only on strictly greater => among equal maxima the largest j wins
                    if (gap <= MAX_GAP && s > 0 && s >= cur.score) { cur.score = s; ptr = j; }
                }
            }
        }
        if (valid) atomicMax(&best[cur.root], best_payload((uint32_t)cur.score, t - ck.a_begin, cur.depth));   // chain.rs:952-964
#pragma unroll
        for (int b = PB - 1; b > 0; b--) prev[b] = prev[b - 1];
        prev[0] = cur;
    }
}

// Thread-per-chunk chaining for small bands (c >= 63): chain_anchors_ani + get_chain_intervals fused.
// A wave chains 64 chunks in lockstep; every lane walks its own chunk sequentially and keeps
//   * the last NB anchors (q, r, ref contig/strand, score, depth | component slot) in REGISTERS as a shift register, so the
//     predecessor scan is a fully unrolled, branch-free block of integer selects;
//   * a table of the LIVE pointer-forest components (those with an anchor still inside the ring -- only they can be extended,
//     chain.rs:859-863), laid out [slot][lane]: the component's argmax record (score | best index | chain length) and
//     root << 8 | reference count.  Up to band+1 components can be live, but more than a handful almost never are: the first
//     DP_LDS_SLOTS (8) slots (the allocator hands out the lowest free slot) sit in LDS, the rest in a global spill table that is
//     practically never touched.  LDS per wave drops from 12(band+1) x 64 B to 6 KB (8 slots), which triples the waves per SIMD.
// All 64 lanes evaluate links (the sweep kernel keeps band/64 of them busy).  When the last anchor of a component leaves the
// ring the component is final and, if it reaches 3 anchors / score 45 (chain.rs:954-977), its interval is emitted straight
// away: the kernel writes nothing per anchor.
// the interval record of a finished chain (root anchor .. best anchor), back in contig-local coordinates (types.rs:508-519)
struct EmitCtx { const uint32_t *anc_q, *anc_r; const PairDesc* pairs; const uint32_t *pc0, *pi0; uint32_t* ivl_cnt; Interval* ivls; uint32_t* err; };
__device__ __forceinline__ bool dp_keep(unsigned long long b) {                  // chain.rs:954-957, 974-977
    const uint32_t sc = (uint32_t)(b >> 40), na = (uint32_t)(b & 0xFFFFFu);
    return na >= MIN_ANCHORS && (int32_t)sc >= MIN_SCORE;
}
// writes the record of a kept chain as the pair's k-th candidate interval
__device__ __forceinline__ void dp_write(const Chunk& ck, uint32_t slot, uint32_t p, uint32_t root, unsigned long long b, const EmitCtx& ec, uint32_t k) {
    const uint32_t sc = (uint32_t)(b >> 40), bi = (uint32_t)((b >> 20) & 0xFFFFFu), na = (uint32_t)(b & 0xFFFFFu);
    if (ec.pi0[p] + k >= ec.pi0[p + 1]) { atomicAdd(ec.err, 1u); return; }
    const uint2 ar = make_uint2(ec.anc_q[ck.a_begin + root], ec.anc_r[ck.a_begin + root]), ab = make_uint2(ec.anc_q[ck.a_begin + bi], ec.anc_r[ck.a_begin + bi]);
    const PairDesc& pd = ec.pairs[p];
    const uint32_t* bo = pd.b_goff;
    const uint32_t ra = ar.y >> 1, rb = ab.y >> 1;
    const uint32_t rctg = ctg_of(bo, pd.b_nctg, ra), roff = bo[rctg];
    Interval iv;
    iv.score = sc; iv.na = na; iv.q0 = ar.x - ck.qoff; iv.q1 = ab.x - ck.qoff;
    iv.r0 = (ra < rb ? ra : rb) - roff; iv.r1 = (ra < rb ? rb : ra) - roff;
    iv.rctg = rctg; iv.qctg = ck.qctg; iv.chunk = slot - ec.pc0[p]; iv.rev = ar.y & 1u;
    ec.ivls[ec.pi0[p] + k] = iv;
}
__device__ __forceinline__ void dp_emit(const Chunk& ck, uint32_t slot, uint32_t p, uint32_t root, unsigned long long b, const EmitCtx& ec) {
    if (dp_keep(b)) dp_write(ck, slot, p, root, b, ec, atomicAdd(&ec.ivl_cnt[p], 1u));
}

// The 64 lanes of a wave step through their chunks in lockstep, so a wave takes as long as its longest chunk: chunks are
// handed out in order of decreasing anchor count (dp_order_keys_kernel + a 10-bit radix sort), which puts chunks of nearly equal
// length side by side (in slot order a wave's lanes are busy only ~1/3 of the time: mean 131 anchors, longest of 64 ~350).
__global__ __launch_bounds__(256) void dp_order_keys_kernel(uint32_t n_slots, const Chunk* chunks, uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    const uint32_t len = chunks[i].a_end - chunks[i].a_begin;
    keys[i] = 1023u - (len >= 1023u ? 1023u : len); vals[i] = i;
}

#ifndef DP_EMIT_Q
#define DP_EMIT_Q 6    // parked chains per chunk (16 B each in a global queue)
#endif
#ifndef DP_LINE
#define DP_LINE 8     // anchors per fetched line: 8 (32 B) measured best (2.04 ms; 16: 2.44 ms, 4: 2.06 ms) -- less LDS, one more wave per SIMD
#endif
template <int NB, int T, uint32_t DP_LDS_SLOTS, bool EXACT>   // EXACT: band == NB (the presets' bands), no per-slot band test
__global__ __launch_bounds__(T) void chain_dp_thread_kernel(uint32_t n_slots, const Chunk* chunks, const uint32_t* chunk_pair, const uint32_t* order, uint32_t band,
                                                            EmitCtx ec, unsigned long long* spill_best, uint32_t* spill_rr, uint4* emit_q, uint32_t emit_cap) {
    __shared__ unsigned long long lds_best[DP_LDS_SLOTS * T];                       // [slot][lane]
    __shared__ uint32_t lds_rr[DP_LDS_SLOTS * T];                                   // [slot][lane]: root << 8 | refcount
    const uint32_t C = band + 1, tid = threadIdx.x;
    const uint32_t thr = blockIdx.x * T + tid;
    const uint32_t slot = thr < n_slots ? order[thr] : n_slots;
    const size_t n_thr = (size_t)gridDim.x * T;                                     // spill tables: [slot - DP_LDS_SLOTS][thread]
    auto get_best = [&](uint32_t c) { return c < DP_LDS_SLOTS ? lds_best[c * T + tid] : spill_best[(size_t)(c - DP_LDS_SLOTS) * n_thr + thr]; };
    auto set_best = [&](uint32_t c, unsigned long long v) { if (c < DP_LDS_SLOTS) lds_best[c * T + tid] = v; else spill_best[(size_t)(c - DP_LDS_SLOTS) * n_thr + thr] = v; };
    auto get_rr = [&](uint32_t c) { return c < DP_LDS_SLOTS ? lds_rr[c * T + tid] : spill_rr[(size_t)(c - DP_LDS_SLOTS) * n_thr + thr]; };
    auto set_rr = [&](uint32_t c, uint32_t v) { if (c < DP_LDS_SLOTS) lds_rr[c * T + tid] = v; else spill_rr[(size_t)(c - DP_LDS_SLOTS) * n_thr + thr] = v; };
    Chunk ck{0, 0, 0, 0, 0, 0};
    if (slot < n_slots) ck = chunks[slot];
    const uint32_t n = ck.a_end - ck.a_begin;
    if (n >= MAX_CHUNK_ANCHORS) { atomicAdd(ec.err, 1u); return; }
    const uint32_t p = n ? chunk_pair[slot] : 0;
    // A finished chain is not turned into its interval record on the spot: that is a chain of dependent global round trips (reserve a slot,
    // fetch two anchors, search the contig table) during which the other 63 lanes of the wave would wait, once for every chain of every lane.
    // The lane parks (root, best) in its column of a global queue -- a store, nothing to wait for -- and all lanes write their records
    // together after the scan.  DP_EMIT_Q chains per chunk fit (mean 2); further ones are written directly.
    uint32_t nq = 0;
    auto emit = [&](uint32_t root, unsigned long long b) {
        if (!dp_keep(b)) return;
        // (the widest rings have no register to spare for the queue: with it the NB = 83 kernel spills)
        if (NB <= 40 && nq < emit_cap) { emit_q[(size_t)nq * n_thr + thr] = make_uint4(root, (uint32_t)b, (uint32_t)(b >> 32), 0u); nq++; }
        else dp_emit(ck, slot, p, root, b, ec);
    };
    unsigned long long free_mask = C >= 64 ? ~0ull : ((1ull << C) - 1ull);
    // ring of the last NB anchors: q + 1, strand-signed r + 1, score + ANCHOR_SCORE, depth << 8 | component.
    //  * r is kept as s = reverse ? ~r : r.  For two anchors of the same strand s_i - s_j is the forward distance on that strand
    //    (chain.rs:573-586); for different strands it is >= 2 * CTG_PAD away from 0 in both directions because every padded
    //    coordinate lies in [CTG_PAD, 2^31 - CTG_PAD) -- the same-contig and the same-strand tests are both implied by the gap test.
    //  * both coordinates are stored + 1, so that the differences come out as dq - 1 and dr - 1: "0 < dq <= band" is ONE unsigned compare,
    //    the gap |dr - dq| is unchanged, and taken as an UNSIGNED absolute difference (v_sad_u32) it also rejects dr <= 0: with
    //    0 <= dq - 1 < 2500 a negative dr - 1 is >= 2^31 as unsigned and the difference far above MAX_GAP.
    //  * empty slots hold 0, which is more than BP_CHAIN_BAND below any real coordinate.
    uint32_t rq[NB], rr[NB], rs[NB], rd[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) { rq[k] = 0; rr[k] = 0; rs[k] = 0; rd[k] = 0; }
    // Anchor fetch.  A lane walks its own chunk, so a plain per-lane load touches 64 different cache lines per instruction and
    // uses 4 bytes of each; with ~50k such streams per XCD the lines are evicted before their next element is wanted and every
    // anchor costs a 64-byte HBM fetch (measured: 15 GB read for 2.4 GB of anchors).  Instead every lane pulls whole lines
    // (DP_LINE anchors of one array) as 16-byte loads, one line ahead of use, and parks the current line in its own LDS
    // column [element][lane].  All lanes use the same element index: a lane's walk starts at its chunk's 64-byte-aligned
    // predecessor ("virtual" index v; elements before the chunk are skipped), which keeps the LDS reads conflict-free and
    // the refill branch wave-uniform.
    constexpr uint32_t LINE = DP_LINE;                       // anchors per fetched line (16 = 64 bytes)
    constexpr int LQ = LINE / 4;
    __shared__ uint32_t lds_q[LINE * T], lds_r[LINE * T];
    const uint32_t voff = ck.a_begin & (LINE - 1);
    const uint32_t vtot = n ? n + voff : 0;
    const uint32_t* line_q = ec.anc_q + (ck.a_begin - voff); const uint32_t* line_r = ec.anc_r + (ck.a_begin - voff);
    uint4 pq[LQ], pr[LQ];
#pragma unroll
    for (int j = 0; j < LQ; j++) { pq[j] = make_uint4(0, 0, 0, 0); pr[j] = make_uint4(0, 0, 0, 0); }
    if (vtot) {
#pragma unroll
        for (int j = 0; j < LQ; j++) { pq[j] = *(const uint4*)(line_q + 4 * j); pr[j] = *(const uint4*)(line_r + 4 * j); }
    }
    for (uint32_t v = 0;; v++) {
        const uint32_t kk = v & (LINE - 1);
        if (kk == 0) {                                                              // wave-uniform
            if (__ballot(v < vtot) == 0) break;
#pragma unroll
            for (int j = 0; j < LQ; j++) {
                lds_q[(4 * j + 0) * T + tid] = pq[j].x; lds_q[(4 * j + 1) * T + tid] = pq[j].y; lds_q[(4 * j + 2) * T + tid] = pq[j].z; lds_q[(4 * j + 3) * T + tid] = pq[j].w;
                lds_r[(4 * j + 0) * T + tid] = pr[j].x; lds_r[(4 * j + 1) * T + tid] = pr[j].y; lds_r[(4 * j + 2) * T + tid] = pr[j].z; lds_r[(4 * j + 3) * T + tid] = pr[j].w;
            }
            if (v + LINE < vtot) {
#pragma unroll
                for (int j = 0; j < LQ; j++) { pq[j] = *(const uint4*)(line_q + v + LINE + 4 * j); pr[j] = *(const uint4*)(line_r + v + LINE + 4 * j); }
            }
        }
        if (v < voff || v >= vtot) continue;
        const uint32_t i = v - voff;
        const uint2 a = make_uint2(lds_q[kk * T + tid], lds_r[kk * T + tid]);
        const uint32_t q = a.x, r = (a.y & 1u) ? ~(a.y >> 1) : (a.y >> 1);
        int32_t bscore = 0; uint32_t bdc = NONE;
        // predecessors j = i-1-k for k = 0..band-1 (downward scan; strict '>' keeps the largest j among equal maxima, chain.rs:852-880).
        // Anchors ascend in q, so once the slot just examined is out of reach for every lane the older ones are too: the scan
        // stops there (checked every four slots; chunks are dealt out by length, so a wave's lanes agree on how far to look).
        bool stop = false;
#pragma unroll
        for (int g = 0; g < NB; g += 4) {
            if (g > 0 && !stop) stop = __ballot((int32_t)(q - rq[g - 1]) < (int32_t)BP_CHAIN_BAND) == 0;   // dq - 1 = -1 (equal q) keeps scanning
            if (!stop) {
#pragma unroll
                for (int k = g; k < g + 4 && k < NB; k++) {
                    if (EXACT || (uint32_t)k < band) {
                        const uint32_t dq1 = q - rq[k], dr1 = r - rr[k];            // dq - 1, dr - 1
                        const uint32_t gap = abs_diff_u32(dr1, dq1);
                        const int32_t sc = (int32_t)(rs[k] - gap);
                        // 0 < dq <= 2500 and gap <= 300 bound dr by 2800 < D_MAX_LIN_LENGTH (chain.rs:856-863, 564-597)
                        const bool ok = (dq1 < BP_CHAIN_BAND) & (gap <= (uint32_t)MAX_GAP) & (sc > bscore);
                        bscore = ok ? sc : bscore; bdc = ok ? rd[k] : bdc;
                    }
                }
            }
        }
        uint32_t comp, depth;
        if (bdc != NONE) {
            comp = bdc & 0xFFu; depth = (bdc >> 8) + 1;
            set_rr(comp, get_rr(comp) + 1);
            const unsigned long long pay = best_payload((uint32_t)bscore, i, depth);
            if (pay > get_best(comp)) set_best(comp, pay);                          // argmax, ties -> largest index (chain.rs:952-964)
        } else {                                                                    // new root: at most `band` components are live, one slot is free
            comp = (uint32_t)__ffsll((long long)free_mask) - 1u; free_mask &= free_mask - 1ull; depth = 1;
            set_rr(comp, (i << 8) | 1u); set_best(comp, best_payload(0, i, 1));
        }
        // anchor i-band (a legal predecessor of anchor i, hence handled after the scan) leaves the ring and releases its
        // component; a component without ring members can never be extended again => it is final
        uint32_t leaving = rd[NB - 1];
        if (!EXACT) {
#pragma unroll
            for (int k = 0; k < NB; k++) leaving = ((uint32_t)k == band - 1) ? rd[k] : leaving;
        }
        if (i >= band) {
            const uint32_t c_old = leaving & 0xFFu;
            const uint32_t v = get_rr(c_old) - 1u;
            set_rr(c_old, v);
            if ((v & 0xFFu) == 0) { emit(v >> 8, get_best(c_old)); free_mask |= 1ull << c_old; }
        }
#pragma unroll
        for (int k = NB - 1; k > 0; k--) { rq[k] = rq[k - 1]; rr[k] = rr[k - 1]; rs[k] = rs[k - 1]; rd[k] = rd[k - 1]; }
        rq[0] = q + 1u; rr[0] = r + 1u; rs[0] = (uint32_t)(bscore + ANCHOR_SCORE); rd[0] = (depth << 8) | comp;
    }
    // chunk end: every component still referenced by the ring is final now
    const uint32_t live = n < band ? n : band;
#pragma unroll
    for (int k = 0; k < NB; k++) {
        if ((uint32_t)k < live) {
            const uint32_t c_old = rd[k] & 0xFFu;
            const uint32_t v = get_rr(c_old) - 1u;
            set_rr(c_old, v);
            if ((v & 0xFFu) == 0) emit(v >> 8, get_best(c_old));
        }
    }
    if (NB <= 40 && nq) {
        const uint32_t k0 = atomicAdd(&ec.ivl_cnt[p], nq);
        for (uint32_t e = 0; e < nq; e++) {
            const uint4 r = emit_q[(size_t)e * n_thr + thr];
            dp_write(ck, slot, p, r.x, ((unsigned long long)r.z << 32) | r.y, ec, k0 + e);
        }
    }
}

// chain.rs:939-1007: one candidate interval per pointer-forest component that reaches 3 anchors / score 45
// One wave per chunk: roots are the anchors whose argmax record is non-zero.
__global__ __launch_bounds__(256) void interval_emit_kernel(uint32_t n_slots, const Chunk* chunks, const unsigned long long* best, const uint32_t* chunk_pair, EmitCtx ec) {
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slot >= n_slots) return;
    const Chunk ck = chunks[slot];
    if (ck.a_end <= ck.a_begin) return;
    const uint32_t p = chunk_pair[slot];
    if (ck.a_end - ck.a_begin >= MAX_CHUNK_ANCHORS) { if (lane_id() == 0) atomicAdd(ec.err, 1u); return; }
    for (uint32_t i = ck.a_begin + lane_id(); i < ck.a_end; i += 64) {
        const unsigned long long b = best[i];
        if (b != 0) dp_emit(ck, slot, p, i - ck.a_begin, b, ec);                    // b == 0: not a root
    }
}

// ------------------------------------------------------------------------------------------------ greedy selection
__device__ __forceinline__ int ivl_cmp(const Interval& a, const Interval& b) {     // derived PartialOrd over the field order
#define SKH_CMP(f) if (a.f != b.f) return a.f < b.f ? -1 : 1;
    SKH_CMP(score) SKH_CMP(na) SKH_CMP(q0) SKH_CMP(q1) SKH_CMP(r0) SKH_CMP(r1) SKH_CMP(rctg) SKH_CMP(qctg) SKH_CMP(chunk) SKH_CMP(rev)
#undef SKH_CMP
    return 0;
}

constexpr uint32_t GREEDY_LDS = 2048;   // sorted-index slots per wave kept in LDS by the fallback kernel
constexpr uint32_t GREEDY_FAST = 1024;  // pairs with at most this many candidate intervals take the all-LDS kernel

// Fast path (n <= GREEDY_FAST candidates): one wave per pair, two waves per workgroup, everything staged in LDS.
//   1. bitonic sort of (key, index) with key = score(24) | anchors(20) | top 20 bits of q0; key ties (rare) fall back to
//      the full tuple comparison -> the reference's descending order (chain.rs:1012);
//   2. greedy acceptance 64 candidates at a time: every lane owns one candidate and sums its overlaps against the
//      accepted list (uniform LDS broadcasts, no reductions); the 64 decisions are then resolved in order, an accepted
//      candidate's interval being broadcast (v_readlane) to the later lanes of the same batch (chain.rs:1017-1095).
//   LDS per wave is 36 B x CAP; the kernel is instantiated for CAP = 256 / 512 / 1024 and a pair runs in the smallest one that
//   holds it, so that typical pairs (a few hundred candidates) leave room for 2-3 waves per SIMD: the greedy loop is a chain
//   of dependent instructions, and other waves are the only thing that can fill its issue slots.
//   Pairs are handed out by decreasing candidate count (greedy_order_keys_kernel + a 16-bit radix sort): the kernel ends when its
//   slowest wave does, so the long ones start first.
// Accepted interval, 32 B (two 16-byte LDS reads), threaded on up to three lists: the accepted intervals of its chunk (query axis) and
// those of the one or two GREEDY_BIN-sized bins of the reference axis it touches (intervals spanning more go on a separate short list)
struct AccIvl { uint32_t rctg, r0, r1, qctg, q0, q1; uint16_t qnext, rnext0, rnext1, cand; };   // cand = the interval's index among the pair's candidates
constexpr uint32_t GREEDY_BIN_SHIFT = 15;       // 32 kb reference bins: a chain interval of a 20 kb chunk touches one or two
constexpr uint32_t GREEDY_BUCKETS = 256;        // list heads per axis (hashed chunk id / hashed (contig, bin)); 1 KB per wave keeps four workgroups of the 512 class on a CU
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
__global__ __launch_bounds__(256) void greedy_order_keys_kernel(uint32_t n_pairs, const uint32_t* ivl_cnt, uint64_t* keys, uint32_t* vals) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t n = ivl_cnt[p];
    keys[p] = 0xFFFFu - (n > 0xFFFFu ? 0xFFFFu : n); vals[p] = p;
}
template <uint32_t CAP>
__global__ __launch_bounds__(128) void greedy_fast_kernel(uint32_t n_pairs, const uint32_t* order, const uint32_t* pi0, const uint32_t* pc0, const uint32_t* ivl_cnt,
                                                          const Interval* ivls, uint32_t* ivl_next, uint32_t* chunk_head, uint32_t* n_accepted) {
    __shared__ uint32_t lds_idx[2][CAP];
    __shared__ __attribute__((aligned(16))) AccIvl lds_acc[2][CAP];   // accepted intervals; the sort keys (8 B each) borrow this space first
    __shared__ __attribute__((aligned(4))) uint16_t lds_qh[2][GREEDY_BUCKETS], lds_rh[2][GREEDY_BUCKETS];   // pairs of heads are exchanged as 32-bit words
    __shared__ uint16_t lds_long[2][GREEDY_LONG];
    const uint32_t wv = threadIdx.x >> 6;
    if (blockIdx.x * 2 + wv >= n_pairs) return;
    const uint32_t p = order[blockIdx.x * 2 + wv];
    const uint32_t l = lane_id();
    const uint32_t I0 = pi0[p];
    uint32_t n = ivl_cnt[p]; const uint32_t cap = pi0[p + 1] - I0; if (n > cap) n = cap;
    if (n > CAP || (CAP > 256 && n <= CAP / 2)) return;                              // another instantiation's (or greedy_kernel's) pair
    if (n == 0) { if (l == 0) n_accepted[p] = 0; return; }
    uint32_t N = 1; while (N < n) N <<= 1;
    unsigned long long* key = (unsigned long long*)lds_acc[wv]; uint32_t* idx = lds_idx[wv];
    const Interval* iv = ivls + I0;
    for (uint32_t i = l; i < N; i += 64) {
        unsigned long long kx = 0; uint32_t ix = NONE;
        if (i < n) { const Interval e = iv[i]; kx = ((unsigned long long)e.score << 40) | ((unsigned long long)(e.na & 0xFFFFFu) << 20) | (e.q0 >> 12); ix = i; }
        key[i] = kx; idx[i] = ix;
    }
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
                if (f == NONE) sw = false; else if (s2 == NONE) sw = true;
                else if (kf != ks) sw = kf > ks; else sw = ivl_cmp(iv[f], iv[s2]) > 0;
                if (sw) { idx[i] = b; idx[x] = a; key[i] = kb; key[x] = ka; }
            }
            wave_sync_mem();
        }
    }
    AccIvl* acc = lds_acc[wv];
    uint16_t* qh = lds_qh[wv]; uint16_t* rh = lds_rh[wv]; uint16_t* lng = lds_long[wv];
    for (uint32_t i = l; i < GREEDY_BUCKETS; i += 64) { qh[i] = 0xFFFFu; rh[i] = 0xFFFFu; }
    wave_sync_mem();
    uint32_t nacc = 0, nlong = 0;
    bool long_overflow = false;                                                     // more than GREEDY_LONG wide intervals: fall back to scanning everything
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t s = base + l;
        const bool have = s < n;
        const uint32_t ci = have ? idx[s] : 0;
        Interval c = iv[ci];
        uint32_t sum_r = 0, sum_q = 0, cnt_r = 0, cnt_q = 0;
        auto add_r = [&](const AccIvl& a) {                                        // chain.rs:1030-1045
            const bool hr = a.rctg == c.rctg && a.r0 < c.r1 && c.r0 < a.r1;
            const uint32_t xr = c.r1 - a.r0, yr = a.r1 - c.r0;
            cnt_r += hr ? 1u : 0u; sum_r += hr ? (xr < yr ? xr : yr) : 0u;
        };
        auto add_q = [&](const AccIvl& a) {                                        // chain.rs:1059-1073
            const bool hq = a.qctg == c.qctg && a.q0 < c.q1 && c.q0 < a.q1;
            const uint32_t xq = c.q1 - a.q0, yq = a.q1 - c.q0;
            cnt_q += hq ? 1u : 0u; sum_q += hq ? (xq < yq ? xq : yq) : 0u;
        };
        if (long_overflow) {
            for (uint32_t a = 0; a < nacc; a++) { const AccIvl e = acc[a]; add_r(e); add_q(e); }   // uniform index: LDS broadcast
        } else {
            // Accepted intervals that can overlap this candidate: on the query axis those of its own chunk (chunks are disjoint ranges of one
            // contig), on the reference axis those sharing a bin with it.  An interval listed in two bins is counted in the bin that holds
            // max(candidate start, interval start), a point of the overlap if there is one.
            for (uint32_t a = qh[c.chunk & (GREEDY_BUCKETS - 1u)]; a != 0xFFFFu;) { const AccIvl e = acc[a]; add_q(e); a = e.qnext; }
            const uint32_t c0 = c.r0 >> GREEDY_BIN_SHIFT, c1 = greedy_last_bin(c.r0, c.r1);
            for (uint32_t x = c0; x <= c1; x++) {
                const uint32_t h = greedy_rhash(c.rctg, x);
                for (uint32_t a = rh[h]; a != 0xFFFFu;) {
                    const AccIvl e = acc[a];
                    const uint32_t e0 = e.r0 >> GREEDY_BIN_SHIFT;
                    const bool first = greedy_rhash(e.rctg, e0) == h;               // which of the interval's (at most two, consecutive) bins hangs on this head
                    const uint32_t eb = first ? e0 : e0 + 1u;
                    if (e.rctg == c.rctg && eb == x && x == (c0 > e0 ? c0 : e0)) add_r(e);
                    a = first ? e.rnext0 : e.rnext1;
                }
            }
            for (uint32_t t = 0; t < nlong; t++) add_r(acc[lng[t]]);
        }
        const uint32_t nb = n - base < 64 ? n - base : 64, nacc0 = nacc;
        for (uint32_t b = 0; b < nb; b++) {
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
                    acc[nacc] = AccIvl{actg, ar0, ar1, aqc, aq0, aq1, (uint16_t)(bchunk & (GREEDY_BUCKETS - 1u)) /* its query-axis list, until it is linked */,
                                       0xFFFFu, 0xFFFFu, (uint16_t)bci};
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
            e->qnext = (uint16_t)greedy_push(qh, e->qnext, nacc0 + l);
            const uint32_t b0 = e->r0 >> GREEDY_BIN_SHIFT, b1 = greedy_last_bin(e->r0, e->r1);
            if (b1 - b0 < 2u) {
                e->rnext0 = (uint16_t)greedy_push(rh, greedy_rhash(e->rctg, b0), nacc0 + l);
                if (b1 > b0) e->rnext1 = (uint16_t)greedy_push(rh, greedy_rhash(e->rctg, b1), nacc0 + l);
            }
        }
        wave_sync_mem();
    }
    // good_non_overlap_intervals[chunk_id].push (chain.rs:1086-1094) for all accepted intervals at once: the per-chunk lists are only ever
    // summed over (chunk_stats_kernel), so their order is free -- and a push from inside the loop above would put a global-memory round trip
    // (read the chunk's head) into every one of the ~400 sequential steps of a pair
    for (uint32_t a = l; a < nacc; a += 64) {
        const uint32_t bci = acc[a].cand;
        const uint32_t slot = pc0[p] + iv[bci].chunk;
        ivl_next[I0 + bci] = atomicExch(&chunk_head[slot], I0 + bci);
    }
    if (l == 0) n_accepted[p] = nacc;
}

// Fallback for pairs with more than GREEDY_FAST candidate intervals: one wave per pair: bitonic-sort the pair's candidate
// interval indices into DESCENDING tuple order (chain.rs:1012), then accept greedily (chain.rs:1017-1095).  An accepted
// interval is flagged in bit 31 of its sorted slot and pushed on its chunk's list.
__global__ __launch_bounds__(256) void greedy_kernel(uint32_t n_pairs, const uint32_t* pi0, const uint32_t* ps0, const uint32_t* pc0, const uint32_t* ivl_cnt,
                                                     const Interval* ivls, uint32_t* sorted_glob, uint32_t* ivl_next, uint32_t* chunk_head, uint32_t* n_accepted) {
    __shared__ uint32_t lds_idx[4][GREEDY_LDS];
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + wv;
    if (p >= n_pairs) return;
    const uint32_t l = lane_id();
    const uint32_t I0 = pi0[p];
    uint32_t n = ivl_cnt[p]; const uint32_t cap = pi0[p + 1] - I0; if (n > cap) n = cap;
    if (n <= GREEDY_FAST) return;                                                   // handled by greedy_fast_kernel
    uint32_t N = 1; while (N < n) N <<= 1;                                          // ps0 reserves pow2(cap) >= N slots per pair
    uint32_t* idx = N <= GREEDY_LDS ? lds_idx[wv] : sorted_glob + ps0[p];
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

// ------------------------------------------------------------------------------------------------ per-chunk ANI inputs
// chain.rs:199-413.  A wave owns 64 consecutive chunks.  Lane j first walks chunk j's accepted intervals (1-3 of them);
// then the wave visits the 64 chunks one after the other: chunk j's interval bounds are broadcast with v_readlane and all
// 64 lanes stream its ~160 query seed positions (the enumerated sketch's position array, masked by the join's "listed" bits) as
// coalesced 256-byte reads, counting the listed positions, those inside the union of the (padded) intervals and those inside
// the covered range with ballots; finally lane j turns chunk j's counts into its ANI estimate and weight.  (A thread-per-chunk walk of the position list touches 64 different cache lines per load and fetched the
// list 4-5 times over.)
constexpr int STATS_REG = 4;   // intervals of one chunk kept in registers (more -> slow path re-walks the list per position)

__global__ __launch_bounds__(256) void chunk_stats_kernel(uint32_t n_slots, const Chunk* chunks, const uint32_t* chunk_pair, const uint32_t* chunk_head,
                                                          const uint32_t* ivl_next, const Interval* ivls, const PairDesc* pairs, const unsigned long long* inq_mask,
                                                          uint32_t c, uint32_t k, double* chunk_est, uint32_t* chunk_w, uint4* chunk_sums) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t l = lane_id();
    const bool valid = slot < n_slots;
    const uint32_t head = valid ? chunk_head[slot] : NONE;
    if (valid) chunk_w[slot] = NONE;                                                // NONE = no estimate from this chunk
    uint32_t total_anchors = 0, rq0 = 0xFFFFFFFFu, rq1 = 0, tbcq = 0, sum_len = 0, n_int = 0, s_begin = 0, s_end = 0, qoff = 0;
    uint32_t lo[STATS_REG], hi[STATS_REG];
#pragma unroll
    for (int i = 0; i < STATS_REG; i++) { lo[i] = 1; hi[i] = 0; }                   // empty
    bool active = false;
    const uint32_t* ag = nullptr; const unsigned long long* mk = nullptr;           // the chunk's pair: position array and "listed" bits
    if (head != NONE) {                                                             // else total_anchors == 0 (chain.rs:253)
        const Chunk ck = chunks[slot];
        const uint32_t p = chunk_pair[slot];
        const bool switched = (pairs[p].flags & 4u) != 0;
        ag = pairs[p].a_g; mk = inq_mask + (uint64_t)pairs[p].tile0 * (JOIN_TILE / 64);
        s_begin = ck.s_begin; s_end = ck.s_end; qoff = ck.qoff;
        for (uint32_t e = head; e != NONE; e = ivl_next[e]) {
            const Interval iv = ivls[e];
            total_anchors += iv.na;
            if (iv.q0 < rq0) rq0 = iv.q0;
            if (iv.q1 > rq1) rq1 = iv.q1;
            tbcq += (switched ? iv.r1 - iv.r0 : iv.q1 - iv.q0) + k + 2 * c;         // chain.rs:223-237
            sum_len += (iv.q1 - iv.q0) + 2 * c + k;                                 // chain.rs:245-249 (overlap is always 0, chain.rs:1091-1093)
            const uint32_t l0 = iv.q0 > c ? iv.q0 - c : 0, h0 = iv.q1 + c;          // chain.rs:239-242
#pragma unroll
            for (int i = 0; i < STATS_REG; i++) if (n_int == (uint32_t)i) { lo[i] = l0; hi[i] = h0; }
            n_int++;
        }
        const bool sensitive = c < 200;                                             // chain.rs:184-190
        active = rq1 - rq0 >= MIN_LENGTH_COVER;                                     // chain.rs:257
        // the chunk's share of the pair totals (summed per pair by finalize_kernel; per-pair atomics from 245 chunks cost more
        // than the rest of this kernel): x = covered-length sum, y = accepted intervals, z = total_query_bases share
        // (chain.rs:184-190: sensitive -> interval lengths, else the chunk's covered range, chain.rs:261-264)
        chunk_sums[slot] = make_uint4(sum_len, n_int, sensitive ? sum_len : (active ? rq1 - rq0 + 2 * c + k : 0u), 0u);
    } else if (valid) chunk_sums[slot] = make_uint4(0, 0, 0, 0);
    uint32_t in_u = 0, in_range = 0, in_list = 0;
    unsigned long long todo = __ballot(active);
    // the first 256 positions of a chunk are fetched as four independent loads, and the next chunk's are in flight while the
    // current chunk is counted: the loop is otherwise a chain of dependent round trips to memory.  A fetched value is
    // coordinate << 1 | listed.  (Assembling the 64 "listed" bits of a block from two wave-uniform loads instead of one load
    // per lane is slower: 1.27 vs 0.79 ms -- the scalar loads sit in the dependent chain.)
    constexpr int PF = 4;
    uint32_t cur[PF], nxt[PF];
    auto fetch = [&](const uint32_t* ag_j, const unsigned long long* mk_j, uint32_t s2, uint32_t se_j) -> uint32_t {
        if (s2 >= se_j) return 0u;
        return (ag_j[s2] & ~1u) | (uint32_t)((mk_j[s2 >> 6] >> (s2 & 63u)) & 1ull);
    };
    auto bcast_ptr = [&](const void* ptr, int src) -> const void* {
        const unsigned long long v = (unsigned long long)ptr;
        const uint32_t lo32 = wave_readlane((uint32_t)v, src), hi32 = wave_readlane((uint32_t)(v >> 32), src);
        return (const void*)(((unsigned long long)hi32 << 32) | lo32);
    };
    int j = -1; uint32_t sb = 0, se = 0;
    const uint32_t* agj = nullptr; const unsigned long long* mkj = nullptr;
    if (todo) {
        j = __ffsll((long long)todo) - 1; todo &= todo - 1ull;
        sb = wave_readlane(s_begin, j); se = wave_readlane(s_end, j);
        agj = (const uint32_t*)bcast_ptr(ag, j); mkj = (const unsigned long long*)bcast_ptr(mk, j);
#pragma unroll
        for (int u = 0; u < PF; u++) cur[u] = fetch(agj, mkj, sb + 64u * (uint32_t)u + l, se);
    }
    while (j >= 0) {                                                                // wave-uniform
        int jn = -1; uint32_t sbn = 0, sen = 0;
        const uint32_t* agn = nullptr; const unsigned long long* mkn = nullptr;
        if (todo) {
            jn = __ffsll((long long)todo) - 1; todo &= todo - 1ull;
            sbn = wave_readlane(s_begin, jn); sen = wave_readlane(s_end, jn);
            agn = (const uint32_t*)bcast_ptr(ag, jn); mkn = (const unsigned long long*)bcast_ptr(mk, jn);
#pragma unroll
            for (int u = 0; u < PF; u++) nxt[u] = fetch(agn, mkn, sbn + 64u * (uint32_t)u + l, sen);
        }
        const uint32_t nj = wave_readlane(n_int, j), q0j = wave_readlane(rq0, j), q1j = wave_readlane(rq1, j), headj = wave_readlane(head, j);
        const uint32_t qoffj = wave_readlane(qoff, j);                              // positions are padded coordinates; intervals are contig-local
        uint32_t lj[STATS_REG], hj[STATS_REG];
#pragma unroll
        for (int i = 0; i < STATS_REG; i++) { lj[i] = wave_readlane(lo[i], j); hj[i] = wave_readlane(hi[i], j); }
        uint32_t cu = 0, cr = 0, cl = 0;
        auto count = [&](uint32_t v) {
            const bool on = (v & 1u) != 0;                                          // listed in query_positions_all (0 beyond the chunk)
            const uint32_t pos = (v >> 1) - qoffj;
            bool hit = false;
            if (nj <= (uint32_t)STATS_REG) {
#pragma unroll
                for (int i = 0; i < STATS_REG; i++) hit = hit || (pos >= lj[i] && pos <= hj[i]);
            } else {
                for (uint32_t e = headj; e != NONE; e = ivl_next[e]) { const Interval iv = ivls[e]; const uint32_t l0 = iv.q0 > c ? iv.q0 - c : 0; hit = hit || (pos >= l0 && pos <= iv.q1 + c); }
            }
            cl += (uint32_t)__popcll(__ballot(on));                                 // chain.rs:755-780: seeds of the chunk
            cu += (uint32_t)__popcll(__ballot(on && hit));                          // chain.rs:268-272
            cr += (uint32_t)__popcll(__ballot(on && pos >= q0j && pos <= q1j));     // chain.rs:326-332 (spacing estimates are 0)
        };
#pragma unroll
        for (int u = 0; u < PF; u++) if (sb + 64u * (uint32_t)u < se) count(cur[u]);
        for (uint32_t b2 = sb + 64u * PF; b2 < se; b2 += 64) count(fetch(agj, mkj, b2 + l, se));
        if ((int)l == j) { in_u = cu; in_range = cr; in_list = cl; }
        j = jn; sb = sbn; se = sen; agj = agn; mkj = mkn;
#pragma unroll
        for (int u = 0; u < PF; u++) cur[u] = nxt[u];
    }
    if (!active) return;
    uint32_t considered = in_list;
    const double inv_k = 1. / (double)k;
    const double putative = pow((double)total_anchors / (double)in_u, inv_k);       // chain.rs:335-339
    if (putative > 0.950 && tbcq > c * 4 && rq1 - rq0 < CHUNK_SIZE * 9 / 10 && (double)considered > 1.05 * (double)in_range)
        considered = in_range;                                                      // chain.rs:340-351
    double ml = (double)total_anchors / (double)considered;
    if (!(ml < 1.)) ml = 1.;                                                        // f64::min(1., x) (x = NaN or >= 1 -> 1)
    chunk_est[slot] = pow(ml, inv_k); chunk_w[slot] = considered;                   // chain.rs:363-396
}

// ------------------------------------------------------------------------------------------------ per-pair result
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct FinalizeArgs {
    uint32_t n_pairs, c, k;
    double min_af, both_min_af; int robust, median, learned, compute_ci;
    const GbdtModel::Node* nodes; const uint32_t* tree_off; uint32_t n_trees; float shrinkage, bias;
};
struct FinalizeScratch { double *u_est, *s_est; uint32_t *u_w, *s_w; uint64_t* cum; };

// fastrand 1.9.0 WyRand stream seeded with 7 (chain.rs:62); draw number d (0-based) is a pure function of d
constexpr uint64_t WYRAND_STEP = 0xA0761D6478BD642Full;
// output for generator state s: low ^ high half of the 128-bit product s * (s ^ c), from four 32x32+64 multiply-adds
__device__ __forceinline__ uint64_t wyrand_mix(uint64_t s) {
    const uint64_t b = s ^ 0xE7037ED1A0B428DBull;
    const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)s0 * b0;
    const uint64_t p01 = (uint64_t)s0 * b1 + (p00 >> 32);
    const uint64_t p10 = (uint64_t)s1 * b0 + (uint32_t)p01;
    const uint64_t p11 = (uint64_t)s1 * b1 + (p01 >> 32) + (p10 >> 32);
    return ((p10 << 32) | (uint32_t)p00) ^ p11;
}
__device__ __forceinline__ uint64_t wyrand_state(uint64_t d) { return 7ull + (d + 1ull) * WYRAND_STEP; }   // state after d + 1 steps from seed 7
__device__ __forceinline__ uint64_t wyrand_draw(uint64_t d) { return wyrand_mix(wyrand_state(d)); }

// chain.rs:414-555 + regression.rs:30-64.  One wave per pair.  The per-pair work arrays (one entry per chunk) live in LDS;
// the kernel is instantiated for FIN_LDS = 320 (genomes up to ~6 Mbp: 11 KB per wave, 3 waves per SIMD) and 1024 entries and
// a pair runs in the smaller one that holds it; beyond 1024 chunks the arrays spill to global scratch.  The kernel is a chain
// of dependent LDS reads, shuffles and f64 arithmetic -- other waves are what fills its issue slots.
template <uint32_t FIN_LDS, uint32_t FIN_MIN>
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs fa, const PairDesc* pairs, const uint32_t* pc0, const uint32_t* n_chunks,
                                                       const double* chunk_est, const uint32_t* chunk_w, const uint4* chunk_sums, FinalizeScratch fs,
                                                       uint32_t* n_est_out, skh_ani_result* out) {
    __shared__ double lds_boot[4][128];
    __shared__ double lds_u[4][FIN_LDS], lds_s[4][FIN_LDS];
    __shared__ uint64_t lds_cum[4][FIN_LDS];
    __shared__ uint32_t lds_uw[4][FIN_LDS], lds_sw[4][FIN_LDS];
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + wv;
    if (p >= fa.n_pairs) return;
    const uint32_t l = lane_id();
    const PairDesc pd = pairs[p];
    const uint32_t C0 = pc0[p], nc = n_chunks[p];
    if (nc < FIN_MIN || (FIN_LDS < 1024 && nc > FIN_LDS)) return;                   // the other instantiation's pair
    const bool in_lds = nc <= FIN_LDS;
    double* U = in_lds ? lds_u[wv] : fs.u_est + C0; uint32_t* UW = in_lds ? lds_uw[wv] : fs.u_w + C0;
    double* S = in_lds ? lds_s[wv] : fs.s_est + C0; uint32_t* SW = in_lds ? lds_sw[wv] : fs.s_w + C0;
    uint64_t* CUM = in_lds ? lds_cum[wv] : fs.cum + C0;
    // 1. valid (estimate, weight) pairs in chunk order
    uint32_t n = 0, acl = 0, nchains = 0, tqb = 0;
    for (uint32_t b = 0; b < nc; b += 64) {
        const uint32_t s = C0 + b + l;
        if (b + l < nc) { const uint4 cs = chunk_sums[s]; acl += cs.x; nchains += cs.y; tqb += cs.z; }
        const bool v = b + l < nc && chunk_w[s] != NONE;
        const unsigned long long m = __ballot(v);
        if (v) { const uint32_t o = n + (uint32_t)__popcll(m & ((1ull << l) - 1ull)); U[o] = chunk_est[s]; UW[o] = chunk_w[s]; }
        n += (uint32_t)__popcll(m);
    }
    if (l == 0) n_est_out[p] = n;
    skh_ani_result res;
    memset(&res, 0, sizeof res);
    acl = wave_sum(acl); nchains = wave_sum(nchains); tqb = wave_sum(tqb);
    if (n == 0 || nchains == 0) {                                                   // chain.rs:416-420: AniEstResult::default() with ani = NaN
        res.ani = __builtin_nanf("");
        if (l == 0) out[p] = res;
        return;
    }
    wave_sync_mem();
    // 2. ascending sort by (estimate, weight) by rank counting (chain.rs:414)
    for (uint32_t i = l; i < n; i += 64) {
        const double e = U[i]; const uint32_t w = UW[i];
        uint32_t rank = 0;
#pragma unroll 4
        for (uint32_t j = 0; j < n; j++) { const double ej = U[j]; const uint32_t wj = UW[j]; rank += (ej < e || (ej == e && (wj < w || (wj == w && j < i)))) ? 1u : 0u; }
        S[rank] = e; SW[rank] = w;
    }
    wave_sync_mem();
    // 3. inclusive cumulative weights
    uint64_t carry = 0;
    for (uint32_t b = 0; b < n; b += 64) {
        const uint32_t i = b + l;
        uint64_t v = i < n ? SW[i] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t t = __shfl_up(v, d, 64); if (l >= (uint32_t)d) v += t; }
        if (i < n) CUM[i] = carry + v;
        carry += __shfl(v, 63, 64);
    }
    const uint64_t total_mult = carry;
    wave_sync_mem();
    // 4. quantile window (chain.rs:426-460)
    double lower = 0., upper = 1.;
    if (fa.median) { lower = 0.499; upper = 0.501; } else if (fa.robust) { lower = 0.10; upper = 0.90; }
    const uint64_t thr_lo = (uint64_t)((double)total_mult * lower), thr_hi = (uint64_t)((double)total_mult * upper);
    uint32_t lower_i = n, upper_i = n;   // first indices reaching the thresholds
    for (uint32_t b = 0; b < n && (lower_i == n || upper_i == n); b += 64) {
        const uint32_t i = b + l;
        const uint64_t cv = i < n ? CUM[i] : 0;
        const unsigned long long mlo = __ballot(i < n && cv >= thr_lo), mhi = __ballot(i < n && cv >= thr_hi);
        if (lower_i == n && mlo) lower_i = b + (uint32_t)__ffsll((long long)mlo) - 1u;
        if (upper_i == n && mhi) upper_i = b + (uint32_t)__ffsll((long long)mhi) - 1u;
    }
    if (lower_i == n) lower_i = 0;
    upper_i = upper_i == n ? n - 1 : upper_i + 1;                                   // chain.rs:444,455-458
    // 5. weighted mean over [lower_i, upper_i) and population std of all estimates (chain.rs:462-471, 39-55)
    double wsum = 0., esum = 0.; uint64_t tm = 0;
    for (uint32_t i = l; i < n; i += 64) {
        const double e = S[i]; esum += e;
        if (i >= lower_i && i < upper_i) { wsum += e * (double)SW[i]; tm += SW[i]; }
    }
    wsum = wave_sum_f64(wsum); esum = wave_sum_f64(esum); tm = wave_sum_u64(tm);
    double final_ani = wsum / (double)tm;
    const double mean = esum / (double)n;
    double var = 0.;
    for (uint32_t i = l; i < n; i += 64) { const double d = mean - S[i]; var += d * d; }
    var = wave_sum_f64(var);
    const double sd = sqrt(var / (double)n);
    // 6. percentile bootstrap (chain.rs:57-86): 100 resamples of n draws from the multiplicity-expanded list
    double ci_lo = 0., ci_hi = 1.;
    if (fa.compute_ci && n >= 10) {
        uint32_t nsteps = 0; while ((1u << nsteps) < n) nsteps++;                  // fixed-length branch-free binary search
        // first i with CUM[i] > x; x < total_mult = CUM[n-1], so the answer is in [0, n-1]
        auto search64 = [&](uint64_t x) { uint32_t lo = 0, hi = n - 1; for (uint32_t st = 0; st < nsteps; st++) { const uint32_t mid = (lo + hi) >> 1; const bool gt = CUM[mid] > x; hi = gt ? mid : hi; lo = gt ? lo : mid + 1; } return lo; };
        if (in_lds && total_mult < 0xFFFFFFFFull) {
            // Fast path (every realistic pair): 32-bit cumulative weights, and a 512-entry directory over the value range
            // (bucket b = x >> sh; entry = first | last candidate << 16, one LDS read) that narrows each search to the one or two entries
            // whose cumulative weight falls into the draw's bucket.  Both live in LDS arrays that are dead after the sort (unsorted
            // weights / estimates; 512 x 4 B fit the smaller instantiation's 320 doubles).
            uint32_t* C32 = UW; uint32_t* T = (uint32_t*)U;
            uint32_t sh = 0; while ((total_mult >> sh) >= 512) sh++;
            const uint32_t nb = (uint32_t)(total_mult >> sh) + 1;                      // x < total_mult  =>  x >> sh < nb <= 512
            for (uint32_t i = l; i < n; i += 64) C32[i] = (uint32_t)CUM[i];
            for (uint32_t b = l; b < nb; b += 64) T[b] = search64((uint64_t)b << sh) | (search64((uint64_t)(b + 1) << sh) << 16);   // past-the-end thresholds give n-1
            wave_sync_mem();
            const uint32_t tot32 = (uint32_t)total_mult;
            // generator states of this lane's draws j = l + 64 u (+ 256 m) of resample `it`: advanced by n steps per resample instead of
            // being recomputed from the draw number (a 64-bit multiply per draw)
            uint64_t st[4];
#pragma unroll
            for (int u = 0; u < 4; u++) st[u] = wyrand_state((uint64_t)l + 64u * (uint32_t)u);
            const uint64_t step_it = (uint64_t)n * WYRAND_STEP, step_256 = 256ull * WYRAND_STEP;
            // resamples in groups of four: the four wave reductions (six dependent shuffle steps each) then run interleaved
            for (uint32_t it0 = 0; it0 < 100; it0 += 4) {
              double sg[4];
#pragma unroll
              for (int g = 0; g < 4; g++) {
                double s = 0.;
                uint64_t sm[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { sm[u] = st[u]; st[u] += step_it; }
                for (uint32_t j0 = l; j0 < n; j0 += 256) {
                    uint32_t x[4], lo[4], hi[4]; bool on[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t j = j0 + 64u * (uint32_t)u;
                        on[u] = j < n;
                        const uint64_t r = wyrand_mix(sm[u]); sm[u] += step_256;
                        // Lemire reduction hi64(r * total) for total < 2^32; its rejection branch has probability total/2^64
                        x[u] = (uint32_t)(((uint64_t)(uint32_t)(r >> 32) * tot32 + __umulhi((uint32_t)r, tot32)) >> 32);
                        const uint32_t tb = T[x[u] >> sh];
                        lo[u] = tb & 0xFFFFu; hi[u] = tb >> 16;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        while (lo[u] < hi[u]) { const uint32_t mid = (lo[u] + hi[u]) >> 1; const bool gt = C32[mid] > x[u]; hi[u] = gt ? mid : hi[u]; lo[u] = gt ? lo[u] : mid + 1; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) if (on[u]) s += S[lo[u]];
                }
                sg[g] = s;
              }
#pragma unroll
              for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
                  for (int g = 0; g < 4; g++) sg[g] += __shfl_xor(sg[g], d, 64);
              }
              if (l == 0) {
#pragma unroll
                  for (int g = 0; g < 4; g++) lds_boot[wv][it0 + g] = sg[g] / (double)n;
              }
            }
        } else {
            for (uint32_t it = 0; it < 100; it++) {
                double s = 0.;
                for (uint32_t j0 = l; j0 < n; j0 += 64) {
                    const uint64_t r = wyrand_draw((uint64_t)it * n + j0);
                    s += S[search64(__umul64hi(r, total_mult))];
                }
                s = wave_sum_f64(s);
                if (l == 0) lds_boot[wv][it] = s / (double)n;
            }
        }
        wave_sync_mem();
        for (uint32_t i = l; i < 100; i += 64) {
            const double e = lds_boot[wv][i]; uint32_t rank = 0;
            for (uint32_t j = 0; j < 100; j++) { const double ej = lds_boot[wv][j]; rank += (ej < e || (ej == e && j < i)) ? 1u : 0u; }
            if (rank == 4) lds_boot[wv][100] = e;
            if (rank == 94) lds_boot[wv][101] = e;
        }
        wave_sync_mem();
        ci_lo = lds_boot[wv][100]; ci_hi = lds_boot[wv][101];
    }
    // 7. aligned fractions, cut-offs, output record (chain.rs:477-554) -- computed redundantly by every lane (wave-uniform)
    double cov_q = (double)tqb / (double)pd.query_total_len; if (!(cov_q < 1.)) cov_q = 1.;
    double cov_r = (double)tqb / (double)pd.ref_total_len; if (!(cov_r < 1.)) cov_r = 1.;   // total_ref_range has the same numerator (chain.rs:245-246)
    const double cutoff = fa.min_af < 0. ? 0.15 : fa.min_af;                        // chain.rs:100-107
    if (fa.both_min_af > 0.0) { if (cov_q < fa.both_min_af || cov_r < fa.both_min_af) final_ani = -1.; }
    else if (cov_q < cutoff && cov_r < cutoff) final_ani = -1.;
    res.ani = (float)final_ani; res.af_query = (float)cov_q; res.af_ref = (float)cov_r;
    res.ci_lower = (float)ci_lo; res.ci_upper = (float)ci_hi; res.std = (float)sd;
    res.q90_q = pd.q90_q; res.q90_r = pd.q90_r; res.q50_q = pd.q50_q; res.q50_r = pd.q50_r; res.q10_q = pd.q10_q; res.q10_r = pd.q10_r;
    res.num_contigs_q = pd.nctg_q; res.num_contigs_r = pd.nctg_r;
    res.avg_chain_int_len = acl / nchains;                                          // chain.rs:421
    res.total_bases_covered = tqb;
    // 8. learned ANI (regression.rs:30-64; gbdt 0.1.1 LAD predict = bias + sum shrinkage * leaf, f32, tree order).
    //    The 195 tree walks are independent: lanes walk trees lane, lane+64, ...; the f32 sum stays sequential in tree order.
    if (fa.learned && res.ani > 0.9f && res.total_bases_covered > REGRESS_CUTOFF) { // wave-uniform condition
        float x[5];
        x[0] = res.ani * 100.f; x[1] = res.std; x[4] = (float)res.avg_chain_int_len;
        if (res.q50_r > res.q50_q) { x[2] = res.q90_r; x[3] = res.q90_q; } else { x[2] = res.q90_q; x[3] = res.q90_r; }
        float* leaf = (float*)lds_boot[wv];                                         // 256 floats
        wave_sync_mem();
        for (uint32_t t = l; t < fa.n_trees && t < 256; t += 64) {
            const GbdtModel::Node* nd = fa.nodes + fa.tree_off[t]; int32_t i = 0;
            while (nd[i].feat >= 0) {
                const int32_t ft = nd[i].feat;
                const float xv = ft == 0 ? x[0] : ft == 1 ? x[1] : ft == 2 ? x[2] : ft == 3 ? x[3] : x[4];
                i = xv < nd[i].thr ? nd[i].left : nd[i].right;
            }
            leaf[t] = nd[i].pred;
        }
        wave_sync_mem();
        if (l == 0) {
            float pred = fa.bias;
            for (uint32_t t = 0; t < fa.n_trees; t++) {
                float lv;
                if (t < 256) lv = leaf[t];
                else { const GbdtModel::Node* nd = fa.nodes + fa.tree_off[t]; int32_t i = 0; while (nd[i].feat >= 0) i = x[nd[i].feat] < nd[i].thr ? nd[i].left : nd[i].right; lv = nd[i].pred; }
                pred += fa.shrinkage * lv;
            }
            if (pred < 100.f) {
                res.ci_upper = (res.ci_upper - res.ani) + pred / 100.f;
                res.ci_lower = (res.ci_lower - res.ani) + pred / 100.f;
                res.ani = pred / 100.f;
            }
        }
    }
    if (l == 0) out[p] = res;
}

// ------------------------------------------------------------------------------------------------ host driver
namespace {

// chain.rs:15-26 switch_qr with the inputs of chain.rs:625-649
bool is_switched(const skh_sketch_set* R, uint32_t r, const skh_sketch_set* Q, uint32_t q) {
    double q_proxy, r_proxy;
    if (Q->total_len[q] > 100000 && R->total_len[r] > 100000) {
        q_proxy = (double)(Q->mk_off[q + 1] - Q->mk_off[q]) * (double)Q->params.c;
        r_proxy = (double)(R->mk_off[r + 1] - R->mk_off[r]) * (double)R->params.c;
    } else { q_proxy = (double)Q->total_len[q]; r_proxy = (double)R->total_len[r]; }
    const double sq = q_proxy * std::min(Q->mean_ctg[q], 300000.), sr = r_proxy * std::min(R->mean_ctg[r], 300000.);
    if (sq == sr) {                                                                 // query_file_name > ref_file_name
        if (!Q->names.empty() && !R->names.empty()) return Q->names[q] > R->names[r];
        return Q->rank[q] > R->rank[r];
    }
    return sq > sr;
}

// same checksum as the oracle's ora_chain_stats.anchor_checksum: (query contig, query pos, ref contig, ref pos, reverse) per anchor
uint64_t fnv_anchors(const std::vector<uint32_t>& anc, const std::vector<uint32_t>& anc_r, size_t a0, size_t a1, const uint32_t* a_go, uint32_t a_n, const uint32_t* b_go, uint32_t b_n) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t x) { h ^= x; h *= 1099511628211ull; };
    for (size_t i = a0; i < a1; i++) {
        const uint32_t gq = anc[i], gr = anc_r[i] >> 1, rev = anc_r[i] & 1u;
        const uint32_t qc = ctg_of(a_go, a_n, gq), rc = ctg_of(b_go, b_n, gr);
        mix(qc); mix(gq - a_go[qc]); mix(rc); mix(gr - b_go[rc]); mix(rev);
    }
    return h;
}

template <class T> T* upload(skh_ctx* ctx, const std::vector<T>& v) {
    T* d = ctx->arena.get<T>(v.size() ? v.size() : 1);
    h2d(d, v.data(), v.size() * sizeof(T), ctx->stream);
    return d;
}

}  // namespace

// slot order for the join kernels: tiles grouped by (key % 8) and interleaved so that slot b (-> XCD b % 8) serves queue b % 8
// Tables for the join kernels, written on the device from per-pair records: tile -> pair, and slot -> tile, where the tiles of
// pairs [p0, p1) are dealt to eight queues by the probed sketch (queue = key % 8) and queue x owns slots x, x+8, x+16, ...
// (-> XCD x).  The host only computes each pair's first position in its queue.
__global__ __launch_bounds__(256) void tile_pair_kernel(uint32_t n_pairs, const PairDesc* pairs, uint32_t* tile_pair) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t nt = (pairs[p].a_n + JOIN_TILE - 1) / JOIN_TILE, t0 = pairs[p].tile0;
    for (uint32_t t = 0; t < nt; t++) tile_pair[t0 + t] = p;
}
__global__ __launch_bounds__(256) void slot_tile_kernel(uint32_t p0, uint32_t p1, const PairDesc* pairs, const uint32_t* queue_pos /* (p1-p0): queue << 29 | first */,
                                                        uint32_t* slot_tile) {
    const uint32_t p = p0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= p1) return;
    const uint32_t nt = (pairs[p].a_n + JOIN_TILE - 1) / JOIN_TILE, t0 = pairs[p].tile0, qp = queue_pos[p - p0];
    const uint32_t x = qp >> 29, first = qp & 0x1FFFFFFFu;
    for (uint32_t t = 0; t < nt; t++) slot_tile[(size_t)(first + t) * 8 + x] = t0 + t;
}
// returns the device slot table for pairs [p0, p1) and its length
static uint32_t* xcd_slots(skh_ctx* ctx, uint32_t p0, uint32_t p1, const std::vector<PairDesc>& pds, const PairDesc* d_pairs_all, const std::vector<uint32_t>& pair_key,
                           unsigned* n_slots) {
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> qp(p1 - p0);
    for (uint32_t p = p0; p < p1; p++) {
        const uint32_t x = pair_key[p] & 7u, nt = (pds[p].a_n + JOIN_TILE - 1) / JOIN_TILE;
        if (cnt[x] + nt >= (1u << 29)) throw Error("too many join tiles in one batch");
        qp[p - p0] = (x << 29) | cnt[x]; cnt[x] += nt;
    }
    uint32_t mx = 0; for (uint32_t v : cnt) mx = std::max(mx, v);
    *n_slots = mx * 8;
    uint32_t* d_slots = ctx->arena.get<uint32_t>((size_t)mx * 8 + 1);
    if (mx == 0) return d_slots;
    dfill(d_slots, 0xFF, (size_t)mx * 8 * 4, ctx->stream);
    uint32_t* d_qp = ctx->arena.get<uint32_t>(qp.size()); h2d(d_qp, qp.data(), qp.size() * 4, ctx->stream);
    SKH_LAUNCH(slot_tile_kernel, (p1 - p0 + 255) / 256, 256, 0, ctx->stream, p0, p1, d_pairs_all, (const uint32_t*)d_qp, d_slots);
    check_launch("slot_tile");
    return d_slots;
}

void chain_pairs(skh_ctx* ctx, const skh_sketch_set* const* Rsets, uint32_t n_rsets, const uint32_t* pair_rset, const skh_sketch_set* Q, const uint32_t* pair_ref,
                 const uint32_t* pair_query, uint64_t n_pairs_all, const skh_map_params& mp, skh_ani_result* out, skh_chain_stats* stats) {
    if (n_pairs_all == 0) return;
    if (n_pairs_all > 0x7FFFFFFFull) throw std::invalid_argument("too many pairs in one call");
    if (n_rsets == 0 || !Rsets[0]) throw std::invalid_argument("no reference sketch set");
    for (uint32_t x = 0; x < n_rsets; x++)
        if (!Rsets[x] || Rsets[x]->params.c != Q->params.c || Rsets[x]->params.k != Q->params.k) throw std::invalid_argument("ref and query sketches were built with different c/k");
    const uint32_t c = Q->params.c, k = Q->params.k;
    const uint32_t band = BP_CHAIN_BAND / c;                                        // chain.rs:111-112 index_chain_band (ref sketch's c)
    if (band > 256) throw std::invalid_argument("c < 10 (chain band > 256) is not supported by the GPU chaining kernel");
    const GbdtModel* model = nullptr;
    if (mp.learned_ani) {
        model = std::abs((int)c - 125) < std::abs((int)c - 200) ? &ctx->model_c125 : &ctx->model_c200;   // regression.rs:15-22
        if (!model->loaded()) throw std::invalid_argument("learned_ani requested but skh_load_models was not called");
    }
    const uint32_t NP = (uint32_t)n_pairs_all;
    StageTrace tr(ctx);
    // ---- pair descriptors and join tiles
    std::vector<PairDesc> pds(NP); uint64_t n_tiles_all = 0;
    std::vector<uint32_t> chunk_bound(NP), pair_key(NP);
    std::vector<const uint32_t*> host_go_a(stats ? NP : 0), host_go_b(stats ? NP : 0);
    for (uint32_t p = 0; p < NP; p++) {
        const uint32_t rs = pair_rset ? pair_rset[p] : 0u;
        if (rs >= n_rsets) throw std::invalid_argument("pair names a reference set that was not passed");
        const skh_sketch_set* R = Rsets[rs];
        const uint32_t r = pair_ref[p], q = pair_query[p];
        if (r >= R->n_genomes || q >= Q->n_genomes) throw std::invalid_argument("pair index out of range");
        PairDesc& pd = pds[p];
        const bool empty = R->ctg_off[r + 1] == R->ctg_off[r] || Q->ctg_off[q + 1] == Q->ctg_off[q];   // chain.rs:618-620
        const bool sw = is_switched(R, r, Q, q);
        const skh_sketch_set* A = sw ? R : Q; const uint32_t ga = sw ? r : q;       // enumerated side (chain.rs:652-660)
        const skh_sketch_set* B = sw ? Q : R; const uint32_t gb = sw ? q : r;
        pd.a_n = empty ? 0 : (uint32_t)(A->pos_off[ga + 1] - A->pos_off[ga]);
        pd.a_seed = A->p_seed.p + A->pos_off[ga]; pd.a_g = A->p_g.p + A->pos_off[ga]; pd.a_cnt = A->p_cnt.p + A->pos_off[ga];
        pd.b_sg = B->s_g.p + B->pos_off[gb]; pd.b_ent = B->ent.p + B->dist_off[gb]; pd.b_dir = B->dir.p + B->dir_off[gb]; pd.b_nbk = B->n_buckets[gb];
        pd.b_bmap = B->bmap.p + B->bmap_off[gb];
        pd.flags = sw ? 4u : 0u;
        pd.tile0 = (uint32_t)n_tiles_all;
        pd.ref_total_len = R->total_len[r]; pd.query_total_len = Q->total_len[q];
        pd.q10_q = Q->q10[q]; pd.q50_q = Q->q50[q]; pd.q90_q = Q->q90[q]; pd.q10_r = R->q10[r]; pd.q50_r = R->q50[r]; pd.q90_r = R->q90[r];
        pd.nctg_q = (uint32_t)(Q->ctg_off[q + 1] - Q->ctg_off[q]); pd.nctg_r = (uint32_t)(R->ctg_off[r + 1] - R->ctg_off[r]);
        pd.a_goff = A->d_goff.p + A->ctg_off[ga] + ga; pd.b_goff = B->d_goff.p + B->ctg_off[gb] + gb;
        pd.a_nctg = (uint32_t)(A->ctg_off[ga + 1] - A->ctg_off[ga]); pd.b_nctg = (uint32_t)(B->ctg_off[gb + 1] - B->ctg_off[gb]);
        if (stats) { host_go_a[p] = A->goff.data() + A->ctg_off[ga] + ga; host_go_b[p] = B->goff.data() + B->ctg_off[gb] + gb; }
        n_tiles_all += (pd.a_n + JOIN_TILE - 1) / JOIN_TILE;
        if (n_tiles_all >= 0xFFFFFFF0ull) throw std::invalid_argument("too many sketch positions in one chain call; split the pair list");
        pair_key[p] = gb + 3u * (B == Q ? n_rsets : rs);                             // tiles probing the same sketch share an XCD
        // chunks per contig <= len/20000 + 2 (every close advances the end point by 20000 inside the contig)
        chunk_bound[p] = (uint32_t)(A->total_len[ga] / CHUNK_SIZE + 2 * (A->ctg_off[ga + 1] - A->ctg_off[ga]) + 2);
    }
    tr.mark("host: pair descriptors");
    const uint32_t NT = (uint32_t)n_tiles_all;
    PairDesc* d_pairs_all = upload(ctx, pds);
    uint32_t* d_tile_pair = ctx->arena.get<uint32_t>((size_t)NT + 1);
    SKH_LAUNCH(tile_pair_kernel, (NP + 255) / 256, 256, 0, ctx->stream, NP, (const PairDesc*)d_pairs_all, d_tile_pair);
    check_launch("tile_pair");
    skh_ani_result* d_out = ctx->arena.get<skh_ani_result>(NP);
    uint32_t* d_err = ctx->arena.get<uint32_t>(1); dzero(d_err, 4, ctx->stream);
    uint32_t* tile_anch = ctx->arena.get<uint32_t>((size_t)NT + 1);
    uint32_t* d_pair_anch = ctx->arena.get<uint32_t>(NP); uint32_t* d_pair_inq = ctx->arena.get<uint32_t>(NP);
    dzero(d_pair_anch, (size_t)NP * 4, ctx->stream); dzero(d_pair_inq, (size_t)NP * 4, ctx->stream);

    const uint64_t ANCH_BUDGET = ctx->tune.chain_anchors;        // anchors per batch (~30 B of scratch each)
    const uint32_t SUPER_TILES = ctx->tune.chain_super_tiles;    // join tiles per count pass (6 KiB of probe records each)
    auto pow2_at_least = [](uint32_t x) { uint32_t n = 1; while (n < x) n <<= 1; return n; };
    std::vector<uint32_t> pair_anch(NP), pair_inq(NP);
    uint32_t sp0 = 0;
    while (sp0 < NP) {
        // ---- super-batch: pairs [sp0, sp1) = tiles [st0, st1); count pass records one probe result per position
        uint32_t sp1 = sp0;
        while (sp1 < NP && (sp1 == sp0 || (sp1 + 1 < NP ? pds[sp1 + 1].tile0 : NT) - pds[sp0].tile0 <= SUPER_TILES)) sp1++;
        const uint32_t st0 = pds[sp0].tile0, st1 = sp1 < NP ? pds[sp1].tile0 : NT, snt = st1 - st0;
        const std::vector<size_t> super_mark = ctx->arena.mark();
        uint32_t* pinfo = ctx->arena.get<uint32_t>((size_t)snt * JOIN_TILE + 1);
        unsigned long long* inq_mask = ctx->arena.get<unsigned long long>((size_t)snt * (JOIN_TILE / 64) + 1);
        // kernels index tiles globally: shift the record arrays so that tile st0 maps to their start
        uint32_t* pis = pinfo - (size_t)st0 * JOIN_TILE; unsigned long long* imk = inq_mask - (size_t)st0 * (JOIN_TILE / 64);
        uint32_t* d_super_slots = nullptr; unsigned n_super_slots = 0;            // reused by the fill pass when the batch is the whole super-batch
        if (snt) {
            uint32_t* d_slots = xcd_slots(ctx, sp0, sp1, pds, d_pairs_all, pair_key, &n_super_slots);
            d_super_slots = d_slots;
            uint32_t bm_words = 0;                                                 // LDS for the largest bitmap of the batch, up to 32 KB
            for (uint32_t p = sp0; p < sp1; p++) bm_words = std::max(bm_words, ((pds[p].b_nbk + 31) / 32 + 3) / 4 * 4);
            if (bm_words > ctx->tune.join_bitmap_words) bm_words = ctx->tune.join_bitmap_words;   // pairs with a larger bitmap probe the directory directly
            SKH_LAUNCH(join_count_kernel, n_super_slots, 256, (size_t)bm_words * 4, ctx->stream, (const PairDesc*)d_pairs_all, (const uint32_t*)d_slots,
                       (const uint32_t*)d_tile_pair, band, tile_anch, d_pair_anch, d_pair_inq, pis, imk, bm_words);
            check_launch("join_count");
        }
        tr.mark("join_count (+slots)");
        d2h(pair_anch.data() + sp0, d_pair_anch + sp0, (size_t)(sp1 - sp0) * 4, ctx->stream);
        d2h(pair_inq.data() + sp0, d_pair_inq + sp0, (size_t)(sp1 - sp0) * 4, ctx->stream);
        tr.mark("d2h pair counts");
        uint32_t p0 = sp0;
        while (p0 < sp1) {
        uint64_t na = 0; uint32_t p1 = p0;
        while (p1 < sp1 && (p1 == p0 || na + pair_anch[p1] <= ANCH_BUDGET)) { na += pair_anch[p1]; p1++; }
        const uint32_t np = p1 - p0;
        const std::vector<size_t> arena_mark = ctx->arena.mark();
        if (na >= 0xFFFFFFF0ull) throw Error("a single genome pair produces more than 2^32 anchors");
        // per-pair prefix arrays (batch-relative)
        std::vector<uint32_t> pa0(np + 1, 0), pc0(np + 1, 0), pi0(np + 1, 0), ps0(np + 1, 0);
        for (uint32_t i = 0; i < np; i++) {
            pa0[i + 1] = pa0[i] + pair_anch[p0 + i];
            pc0[i + 1] = pc0[i] + (pair_anch[p0 + i] ? std::min(chunk_bound[p0 + i], pair_anch[p0 + i]) : 0);
            const uint32_t icap = pair_anch[p0 + i] / MIN_ANCHORS;
            pi0[i + 1] = pi0[i] + icap; ps0[i + 1] = ps0[i] + (icap > GREEDY_LDS ? pow2_at_least(icap) : 0);   // fallback kernel's global sort scratch
        }
        const uint32_t NA = pa0[np], NC = pc0[np], NI = pi0[np], NS = ps0[np];
        const uint32_t t0 = pds[p0].tile0, t1 = p1 < NP ? pds[p1].tile0 : NT, nt = t1 - t0;
        const PairDesc* d_pairs = d_pairs_all + p0;
        uint32_t* d_pa0 = upload(ctx, pa0); uint32_t* d_pc0 = upload(ctx, pc0);
        uint32_t* d_pi0 = upload(ctx, pi0); uint32_t* d_ps0 = upload(ctx, ps0);
        uint32_t* toff_a = ctx->arena.get<uint32_t>(nt + 1);
        exclusive_scan_u32(ctx, tile_anch + t0, nt, toff_a);
        tr.mark("host prefix + uploads + scans");
        uint32_t* anc_q = ctx->arena.get<uint32_t>((size_t)NA + 16); uint32_t* anc_r = ctx->arena.get<uint32_t>((size_t)NA + 16);
        if (nt) {
            uint32_t* d_slots = d_super_slots; unsigned n_slots = n_super_slots;
            if (t0 != st0 || t1 != st1) d_slots = xcd_slots(ctx, p0, p1, pds, d_pairs_all, pair_key, &n_slots);
            SKH_LAUNCH(join_fill_kernel, n_slots, 256, 0, ctx->stream, (const PairDesc*)d_pairs_all, (const uint32_t*)d_slots,
                       (const uint32_t*)d_tile_pair, t0, (const uint32_t*)toff_a, (const uint32_t*)pis, anc_q, anc_r);
            check_launch("join_fill");
        }
        tr.mark("join_fill (+slots)");
        Chunk* chunks = ctx->arena.get<Chunk>(NC + 1); uint32_t* chunk_pair = ctx->arena.get<uint32_t>(NC + 1);
        uint32_t* n_chunks = ctx->arena.get<uint32_t>(np);
        SKH_LAUNCH(chunk_kernel, (np + 3) / 4, 256, 0, ctx->stream, np, d_pairs, (const uint32_t*)d_pa0,
                   (const uint32_t*)d_pc0, (const uint32_t*)anc_q, chunks, chunk_pair, n_chunks, d_err);
        check_launch("chunk");
        tr.mark("chunk");
        Interval* ivls = ctx->arena.get<Interval>(NI + 1); uint32_t* ivl_cnt = ctx->arena.get<uint32_t>(np);
        uint32_t* ivl_next = ctx->arena.get<uint32_t>(NI + 1); uint32_t* sorted_glob = ctx->arena.get<uint32_t>(NS + 1);
        uint32_t* chunk_head = ctx->arena.get<uint32_t>(NC + 1); uint32_t* n_acc = ctx->arena.get<uint32_t>(np);
        dzero(ivl_cnt, np * 4, ctx->stream); dfill(chunk_head, 0xFF, ((uint64_t)NC + 1) * 4, ctx->stream);
        const EmitCtx ec{anc_q, anc_r, d_pairs, d_pc0, d_pi0, ivl_cnt, ivls, d_err};
        if (NC) {
            if (band <= 84) {   // fused thread-per-chunk chaining + interval emission
                constexpr int T = 64;
                const unsigned gt = (NC + T - 1) / T;
                const uint32_t ls = ctx->tune.chain_dp_lds_slots <= 1 ? 1u : 8u;  // 1: tests push every second live chain through the spill table
                const size_t n_spill = band + 1 > ls ? (size_t)(band + 1 - ls) * gt * T : 1;   // written only by chunks with more live chains than LDS slots
                unsigned long long* spill_best = ctx->arena.get<unsigned long long>(n_spill); uint32_t* spill_rr = ctx->arena.get<uint32_t>(n_spill);
                uint4* emit_q = ctx->arena.get<uint4>((size_t)DP_EMIT_Q * gt * T);
                uint64_t* okeys = ctx->arena.get<uint64_t>(NC); uint32_t* order = ctx->arena.get<uint32_t>(NC);
                SKH_LAUNCH(dp_order_keys_kernel, (NC + 255) / 256, 256, 0, ctx->stream, NC, (const Chunk*)chunks, okeys, order);
                check_launch("dp_order_keys");
                sort_pairs_u64_u32(ctx, okeys, order, NC, 10);
#define SKH_DPT2(NB, LS, EX) SKH_LAUNCH((chain_dp_thread_kernel<NB, T, LS, EX>), gt, T, 0, ctx->stream, NC, (const Chunk*)chunks, (const uint32_t*)chunk_pair, (const uint32_t*)order, band, ec, spill_best, spill_rr, emit_q, ls == 1 ? 1u : (uint32_t)DP_EMIT_Q)   /* test mode: queue of one, the rest written directly */
#define SKH_DPT(NB, EX) do { if (ls == 1) SKH_DPT2(NB, 1, EX); else SKH_DPT2(NB, 8, EX); } while (0)
                // the presets' bands (2500 / c for c = 200, 125, 70, 30) get kernels with exactly that many ring slots
                if (band == 12) SKH_DPT(12, true); else if (band == 20) SKH_DPT(20, true); else if (band == 35) SKH_DPT(35, true); else if (band == 83) SKH_DPT(83, true);
                else if (band <= 12) SKH_DPT(12, false); else if (band <= 20) SKH_DPT(20, false); else if (band <= 28) SKH_DPT(28, false); else if (band <= 40) SKH_DPT(40, false); else SKH_DPT(84, false);
#undef SKH_DPT
#undef SKH_DPT2
                check_launch("chain_dp_thread");
            } else {            // wave-per-chunk sweep + per-anchor argmax records + emit
                unsigned long long* best = ctx->arena.get<unsigned long long>((size_t)NA + 64);
                dzero(best, ((uint64_t)NA + 64) * 8, ctx->stream);
                const unsigned gb = (NC + 3) / 4;
#define SKH_DP(PB) SKH_LAUNCH(chain_dp_kernel<PB>, gb, 256, 0, ctx->stream, NC, (const Chunk*)chunks, band, (const uint32_t*)anc_q, (const uint32_t*)anc_r, best)
                if (band <= 64) SKH_DP(1); else if (band <= 128) SKH_DP(2); else if (band <= 192) SKH_DP(3); else SKH_DP(4);
#undef SKH_DP
                check_launch("chain_dp");
                SKH_LAUNCH(interval_emit_kernel, (NC + 3) / 4, 256, 0, ctx->stream, NC, (const Chunk*)chunks, (const unsigned long long*)best,
                           (const uint32_t*)chunk_pair, ec);
                check_launch("interval_emit");
            }
        }
#define SKH_GREEDY(CAP) SKH_LAUNCH(greedy_fast_kernel<CAP>, (np + 1) / 2, 128, 0, ctx->stream, np, (const uint32_t*)g_order, (const uint32_t*)d_pi0, (const uint32_t*)d_pc0, \
                   (const uint32_t*)ivl_cnt, (const Interval*)ivls, ivl_next, chunk_head, n_acc); check_launch("greedy_fast")
        tr.mark("dp (+order sort)");
        uint64_t* g_keys = ctx->arena.get<uint64_t>(np); uint32_t* g_order = ctx->arena.get<uint32_t>(np);
        SKH_LAUNCH(greedy_order_keys_kernel, (np + 255) / 256, 256, 0, ctx->stream, np, (const uint32_t*)ivl_cnt, g_keys, g_order);
        check_launch("greedy_order_keys");
        sort_pairs_u64_u32(ctx, g_keys, g_order, np, 16);
        SKH_GREEDY(256); SKH_GREEDY(512); SKH_GREEDY(1024);
#undef SKH_GREEDY
        SKH_LAUNCH(greedy_kernel, (np + 3) / 4, 256, 0, ctx->stream, np, (const uint32_t*)d_pi0, (const uint32_t*)d_ps0, (const uint32_t*)d_pc0,
                   (const uint32_t*)ivl_cnt, (const Interval*)ivls, sorted_glob, ivl_next, chunk_head, n_acc);
        check_launch("greedy");
        tr.mark("greedy");
        double* chunk_est = ctx->arena.get<double>(NC + 1); uint32_t* chunk_w = ctx->arena.get<uint32_t>(NC + 1);
        uint4* chunk_sums = ctx->arena.get<uint4>(NC + 1);
        if (NC) {
            SKH_LAUNCH(chunk_stats_kernel, (NC + 255) / 256, 256, 0, ctx->stream, NC, (const Chunk*)chunks, (const uint32_t*)chunk_pair, (const uint32_t*)chunk_head,
                       (const uint32_t*)ivl_next, (const Interval*)ivls, d_pairs, (const unsigned long long*)imk, c, k, chunk_est, chunk_w, chunk_sums);
            check_launch("chunk_stats");
        }
        tr.mark("chunk_stats");
        FinalizeArgs fa{};
        fa.n_pairs = np; fa.c = c; fa.k = k; fa.min_af = mp.min_af; fa.both_min_af = mp.both_min_af; fa.robust = mp.robust; fa.median = mp.median;
        fa.learned = model ? 1 : 0; fa.compute_ci = mp.compute_ci;
        if (model) { fa.nodes = model->nodes.p; fa.tree_off = model->off.p; fa.n_trees = model->n_trees; fa.shrinkage = model->shrinkage; fa.bias = model->bias; }
        FinalizeScratch fs{ctx->arena.get<double>(NC + 1), ctx->arena.get<double>(NC + 1), ctx->arena.get<uint32_t>(NC + 1), ctx->arena.get<uint32_t>(NC + 1),
                           ctx->arena.get<uint64_t>(NC + 1)};
        uint32_t* n_est = ctx->arena.get<uint32_t>(np);
#define SKH_FIN(CAP, MIN) SKH_LAUNCH((finalize_kernel<CAP, MIN>), (np + 3) / 4, 256, 0, ctx->stream, fa, d_pairs, (const uint32_t*)d_pc0, (const uint32_t*)n_chunks, \
                   (const double*)chunk_est, (const uint32_t*)chunk_w, (const uint4*)chunk_sums, fs, n_est, d_out + p0); \
        check_launch("finalize")
        SKH_FIN(320, 0); SKH_FIN(1024, 321);
#undef SKH_FIN
        tr.mark("finalize");
        if (stats) {   // parity/debug path: pull the stage sizes (and the anchors, for the checksum) back to the host
            std::vector<uint32_t> h_nc(np), h_ni(np), h_nacc(np), h_ne(np);
            d2h(h_nc.data(), n_chunks, np * 4, ctx->stream); d2h(h_ni.data(), ivl_cnt, np * 4, ctx->stream);
            d2h(h_nacc.data(), n_acc, np * 4, ctx->stream); d2h(h_ne.data(), n_est, np * 4, ctx->stream);
            std::vector<uint32_t> hanc((size_t)NA), hanr((size_t)NA);
            d2h(hanc.data(), anc_q, (uint64_t)NA * 4, ctx->stream); d2h(hanr.data(), anc_r, (uint64_t)NA * 4, ctx->stream);
            for (uint32_t i = 0; i < np; i++) {
                skh_chain_stats& st = stats[p0 + i];
                st.switched = (pds[p0 + i].flags >> 2) & 1u; st.n_chunks = h_nc[i]; st.n_intervals = h_ni[i]; st.n_accepted = h_nacc[i]; st.n_estimates = h_ne[i];
                st.reserved = 0; st.n_anchors = pair_anch[p0 + i]; st.n_qpos = pair_inq[p0 + i];
                st.anchor_checksum = pair_anch[p0 + i] ? fnv_anchors(hanc, hanr, pa0[i], pa0[i + 1], host_go_a[p0 + i], pds[p0 + i].a_nctg, host_go_b[p0 + i], pds[p0 + i].b_nctg) : 0;
                if (pair_anch[p0 + i] == 0) { st.switched = 1; st.n_qpos = 0; }   // reference returns (default, true) when there are no anchors (chain.rs:619,719)
            }
        }
        dsync(ctx->stream);
        ctx->arena.rewind(arena_mark);
        p0 = p1;
        }
        ctx->arena.rewind(super_mark);
        sp0 = sp1;
    }
    uint32_t h_err = 0;
    d2h(&h_err, d_err, 4, ctx->stream);
    d2h(out, d_out, (uint64_t)NP * sizeof(skh_ani_result), ctx->stream);
    tr.mark("results d2h");
    if (h_err) throw Error("internal capacity bound violated in chain pipeline (" + std::to_string(h_err) + " events)");
}

}  // namespace skh
