// chain_chunk.h -- chunking (chain.rs:738-836): chunk_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

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
// Two-level search: every CHUNK_SAMPLE-th key of the pair's anchor / position arrays is copied to LDS once; a search first narrows its
// range [lo, hi) to one sample interval there (LDS round trips) and only the last log2(CHUNK_SAMPLE) probes go to memory.
// UPPER: first index whose key is > v; otherwise first index whose key is >= v.  samp[t] = key(array[org + t * CHUNK_SAMPLE]), t < ns.
constexpr uint32_t CHUNK_SAMPLE = 128, CHUNK_SAMPLES = 512;     // 2 x 2 KB of LDS per wave; arrays beyond 65,536 entries are searched directly
template <bool UPPER, class Co>
__device__ __forceinline__ void narrow_by_samples(const Co* samp, uint32_t ns, uint32_t org, Co v, uint32_t& lo, uint32_t& hi) {
    if (lo >= hi) return;
    const uint32_t t0 = (lo - org + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE;
    uint32_t t1 = (hi - org + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE; if (t1 > ns) t1 = ns;
    uint32_t a = t0, b = t1;                                                        // first sample in [t0, t1) for which the predicate holds
    while (a < b) { const uint32_t m = (a + b) >> 1; const Co x = samp[m]; if (UPPER ? x > v : x >= v) b = m; else a = m + 1; }
    if (a > t0) { const uint32_t f = org + (a - 1) * CHUNK_SAMPLE + 1; lo = f > lo ? f : lo; }     // sample a-1 fails: the answer lies beyond it
    if (a < t1) { const uint32_t t = org + a * CHUNK_SAMPLE; hi = t < hi ? t : hi; }               // sample a holds: the answer is at or before it
}

// N binary searches of a lane advanced TOGETHER, four ways per step: every step probes the three quarter points of each open range at once (independent
// loads), the last step reads the remaining (up to eight) entries at once -- a range of 128 closes in three dependent round trips to memory instead of
// seven.  The kernel is a chain of such round trips (a wave per pair, 64 searches wide): their number is what it costs.
// arr[x][i] >> sh[x] is the key of entry i; the answer for search x is the first index in [lo, hi) whose key is > v[x] (upper[x]) or >= v[x], else hi.
// (Arr: const uint32_t*, or CoArr in a Wide run -- chain_types.h; Co: the key type)
template <int N, class Arr, class Co>
__device__ __forceinline__ void search_together(const Arr (&arr)[N], const uint32_t (&sh)[N], uint32_t (&lo)[N], uint32_t (&hi)[N], const Co (&v)[N],
                                                const bool (&upper)[N]) {
    auto holds = [&](int x, Co key) { return upper[x] ? key > v[x] : key >= v[x]; };
    for (;;) {
        bool open = false;
#pragma unroll
        for (int x = 0; x < N; x++) open = open || lo[x] < hi[x];
        if (__ballot(open) == 0ull) break;
        Co k[N][8];
#pragma unroll
        for (int x = 0; x < N; x++) {
            const uint32_t n = hi[x] - lo[x];
            if (lo[x] >= hi[x]) continue;
            if (n <= 8u) {
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) k[x][j] = j < n ? (Co)(arr[x][lo[x] + j] >> sh[x]) : (Co)0;
            } else {
                const uint32_t q = n >> 2;
#pragma unroll
                for (uint32_t j = 1; j < 4; j++) k[x][j] = (Co)(arr[x][lo[x] + j * q] >> sh[x]);
            }
        }
#pragma unroll
        for (int x = 0; x < N; x++) {
            const uint32_t n = hi[x] - lo[x];
            if (lo[x] >= hi[x]) continue;
            if (n <= 8u) {
                uint32_t ans = hi[x];
                for (uint32_t j = n; j-- > 0;) if (holds(x, k[x][j])) ans = lo[x] + j;
                lo[x] = hi[x] = ans;                                                 // closed: lo == hi == the answer
            } else {
                const uint32_t q = n >> 2, p1 = lo[x] + q, p2 = lo[x] + 2 * q, p3 = lo[x] + 3 * q;
                if (holds(x, k[x][1])) hi[x] = p1;
                else if (holds(x, k[x][2])) { lo[x] = p1 + 1; hi[x] = p2; }
                else if (holds(x, k[x][3])) { lo[x] = p2 + 1; hi[x] = p3; }
                else lo[x] = p3 + 1;
            }
        }
    }
}

// (A workgroup of four waves per pair -- 256 boundary searches per round, the running maximum left to one wave -- was measured in round 3: 0.54 instead of
//  0.45 ms.  Three of a pair's four waves then sit at barriers most of the time and take the residency of three other pairs.)
template <class W>
__global__ __launch_bounds__(256) void chunk_kernel(uint32_t n_pairs, const PairDesc* pairs, const WidePair* wide, const uint32_t* pa0, const uint32_t* pan,
                                                    const uint32_t* pc0, const typename W::Co* anc_q,
                                                    Chunk* chunks, uint32_t* chunk_pair, uint32_t* n_chunks, uint32_t* err) {
    using Co = typename W::Co; using Arr = typename W::Arr;
    constexpr Co CO_MAX = ~(Co)0;
    __shared__ Co lds_samp[4][2][CHUNK_SAMPLES];
    Arr anc_arr;                                                                    // the anchors' query coordinates as one of the searched arrays
    if constexpr (W::wide) anc_arr = CoArr{anc_q, 1u}; else anc_arr = global_of(anc_q);
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= n_pairs) return;
    const uint32_t l = lane_id();
    // the pair's anchors: pan[p] of them from pa0[p] on (its stretch of the batch arrays may be longer; a pair that overflowed its stretch is
    // re-run by the host, here it is merely kept inside it)
    const uint32_t A0 = pa0[p], A1 = A0 + (pan[p] < pa0[p + 1] - A0 ? pan[p] : pa0[p + 1] - A0), C0 = pc0[p], C1 = pc0[p + 1];
    uint32_t nc = 0;
    if (A1 > A0) {
        const Arr go = W::a_goff(pairs[p], wide, p);
        const uint32_t nctg = pairs[p].a_nctg;
        const Arr ag = W::a_g(pairs[p], wide, p); const uint32_t Q1 = pairs[p].a_n;   // the enumerated sketch's positions
        const Co q_pair_last = anc_q[A1 - 1];
        const uint32_t ns_a = (A1 - A0 + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE, ns_s = (Q1 + CHUNK_SAMPLE - 1) / CHUNK_SAMPLE;
        const bool sampled = ns_a <= CHUNK_SAMPLES && ns_s <= CHUNK_SAMPLES;
        Co* sa = lds_samp[threadIdx.x >> 6][0]; Co* ss = lds_samp[threadIdx.x >> 6][1];
        if (sampled) {                                                              // (all loads of the samples in flight before the first store: one round trip, not one per 64 samples)
            constexpr uint32_t PER = CHUNK_SAMPLES / 64;
            Co ra[PER], rs[PER];
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) { const uint32_t t = l + 64u * u; ra[u] = t < ns_a ? anc_q[A0 + t * CHUNK_SAMPLE] : (Co)0; rs[u] = t < ns_s ? (Co)(ag[t * CHUNK_SAMPLE] >> 1) : (Co)0; }
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) { const uint32_t t = l + 64u * u; if (t < ns_a) sa[t] = ra[u]; if (t < ns_s) ss[t] = rs[u]; }
            wave_sync_mem();
        }
        uint32_t sf_lo = 0, sf_hi = Q1;
        if (sampled) narrow_by_samples<true>(ss, ns_s, 0, q_pair_last, sf_lo, sf_hi);
        uint32_t s_final = 0;                                                       // the pair's final chunk ends its seed range here (chain.rs:794-824): searched beside the first contigs' searches
        // 64 query contigs per round, one per lane: the contig's anchor range [ca, ce), its first position rc0 and its number of end points;
        // then the (contig, k) items of the round are worked off 64 at a time -- a genome in a thousand contigs costs rounds of searches by the
        // sixty-fourth of its contigs, not by the contig.  (Round 4 measured FOUR items per lane, their eight searches advancing together -- a third of the
        // dependent round trips: 0.78 instead of 0.45 ms, the search state of eight ranges does not stay in registers (scratch), so every step pays for it.)
        uint32_t carry_cid = NONE, carry_t = 0, carry_s = 0; int32_t carry_uu = 0;
        for (uint32_t c0 = 0; c0 < nctg; c0 += 64) {
            const uint32_t cl = c0 + l; const bool cv = cl < nctg;
            const Co cstart = cv ? (Co)go[cl] : CO_MAX, cnext = cv ? (Co)go[cl + 1] : CO_MAX;
            uint32_t lo_a = A0, hi_a = cv ? A1 : A0, lo_e = A0, hi_e = cv ? A1 : A0, lo_r = 0, hi_r = cv ? Q1 : 0;
            if (sampled) {
                narrow_by_samples<false>(sa, ns_a, A0, cstart, lo_a, hi_a); narrow_by_samples<false>(sa, ns_a, A0, cnext, lo_e, hi_e);
                narrow_by_samples<false>(ss, ns_s, 0, cstart, lo_r, hi_r);
            }
            {                                                                       // the searches advance together: their round trips overlap
                const Arr arr[4] = {anc_arr, anc_arr, ag, ag}; const uint32_t sh[4] = {0, 0, 1, 1}; const Co vv[4] = {cstart, cnext, cstart, q_pair_last}; const bool up[4] = {false, false, false, true};
                uint32_t lo4[4] = {lo_a, lo_e, lo_r, c0 == 0 ? sf_lo : 0u}, hi4[4] = {hi_a, hi_e, hi_r, c0 == 0 ? sf_hi : 0u};
                search_together<4, Arr, Co>(arr, sh, lo4, hi4, vv, up);
                lo_a = lo4[0]; lo_e = lo4[1]; lo_r = lo4[2]; if (c0 == 0) s_final = lo4[3];
            }
            const uint32_t ca = lo_a, ce = lo_e, rc0 = lo_r;                        // running_counter = 0 within the contig starts at rc0 (chain.rs:742-744)
            const bool has = cv && ce > ca;
            const Co q_first = has ? anc_q[ca] : (Co)0, q_last = has ? anc_q[ce - 1] : (Co)0;
            const uint32_t kmax = has ? (uint32_t)((q_last - q_first) / CHUNK_SIZE) + 1u : 0u;  // lim_k reaches the contig's last anchor no later than this
            const uint32_t P = wave_incl_scan(kmax), M = __shfl(P, 63, 64);
            for (uint32_t j0 = 0; j0 < M; j0 += 64) {
                const uint32_t j = j0 + l; const bool iv = j < M;
                uint32_t slo = 0, shi = 63;                                        // the lane (contig) that owns item j: first with P > j
#pragma unroll
                for (int st = 0; st < 6; st++) { const uint32_t mid = (slo + shi) >> 1; const uint32_t pm = __shfl(P, (int)mid, 64); if (pm > j) shi = mid; else slo = mid + 1; }
                const int src = (int)(iv ? slo : 63u);
                const uint32_t o_kmax = __shfl(kmax, src, 64), o_P = __shfl(P, src, 64), a_c = __shfl(ca, src, 64), e_c = __shfl(ce, src, 64), r_c = __shfl(rc0, src, 64);
                const Co qf = __shfl(q_first, src, 64), cn = __shfl(cnext, src, 64), cs = __shfl(cstart, src, 64);
                const uint32_t k = j - (o_P - o_kmax) + 1u;
                const uint64_t end64 = (uint64_t)qf + (uint64_t)k * CHUNK_SIZE;
                const Co lim = end64 < (uint64_t)(cn - 1) ? (Co)end64 : cn - 1;   // beyond it: another contig, or past the window
                //   b  = first anchor beyond lim (searching all of the pair's later anchors gives the same answer as searching the contig,
                //        because the contig's successor already lies beyond lim);  sb = first position beyond lim = seed list boundary after chunk k
                uint32_t lo_b = a_c, hi_b = iv ? A1 : a_c, lo_s = 0, hi_s = iv ? Q1 : 0;
                if (sampled) { narrow_by_samples<true>(sa, ns_a, A0, lim, lo_b, hi_b); narrow_by_samples<true>(ss, ns_s, 0, lim, lo_s, hi_s); }
                {
                    const Arr arr[2] = {anc_arr, ag}; const uint32_t sh[2] = {0, 1}; const Co vv[2] = {lim, lim}; const bool up[2] = {true, true};
                    uint32_t lo2[2] = {lo_b, lo_s}, hi2[2] = {hi_b, hi_s};
                    search_together<2, Arr, Co>(arr, sh, lo2, hi2, vv, up);
                    lo_b = lo2[0]; lo_s = lo2[1];
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
                Chunk ck; ck.a_begin = t_prev; ck.a_end = t < e_c ? t : e_c; ck.s_begin = s_prev; ck.s_end = sb; ck.qoff = (uint32_t)cs; ck.qctg = c0 + (uint32_t)src;
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
