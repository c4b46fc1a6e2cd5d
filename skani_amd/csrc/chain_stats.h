// chain_stats.h -- calculate_ani, per chunk (chain.rs:173-396): chunk_stats_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ per-chunk ANI inputs
// chain.rs:199-413.  A wave owns 64 consecutive chunks.  Lane j first walks chunk j's accepted intervals (1-3 of them);
// then the wave visits the 64 chunks one after the other: chunk j's interval bounds are broadcast with v_readlane and all
// 64 lanes stream its ~160 query seed positions (the enumerated sketch's position array, masked by the join's "listed" bits) as
// coalesced 256-byte reads, counting the listed positions, those inside the union of the (padded) intervals and those inside
// the covered range with ballots; finally lane j turns chunk j's counts into its ANI estimate and weight.  (A thread-per-chunk walk of the position list touches 64 different cache lines per load and fetched the
// list 4-5 times over.)
constexpr int STATS_REG = 4;   // intervals of one chunk kept in registers (more -> slow path re-walks the list per position)

template <class W>
__global__ __launch_bounds__(256) void chunk_stats_kernel(uint32_t n_slots, const Chunk* chunks, const uint32_t* chunk_pair, const uint32_t* chunk_head,
                                                          const uint32_t* ivl_next, const Interval* ivls, const PairDesc* pairs, const WidePair* wide, const unsigned long long* inq_mask,
                                                          uint32_t c, uint32_t k, double* chunk_est, uint32_t* chunk_w, uint4* chunk_sums) {
    using Co = typename W::Co; using Arr = typename W::Arr;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t l = lane_id();
    const bool valid = slot < n_slots;
    const uint32_t head = valid ? chunk_head[slot] : NONE;
    if (valid) chunk_w[slot] = NONE;                                                // NONE = no estimate from this chunk
    uint32_t total_anchors = 0, rq0 = 0xFFFFFFFFu, rq1 = 0, tbcq = 0, sum_len = 0, n_int = 0, s_begin = 0, s_end = 0; Co qoff = 0;
    uint32_t lo[STATS_REG], hi[STATS_REG];
#pragma unroll
    for (int i = 0; i < STATS_REG; i++) { lo[i] = 1; hi[i] = 0; }                   // empty
    bool active = false;
    const void* ag = nullptr; uint32_t ag64 = 0; const unsigned long long* mk = nullptr;   // the chunk's pair: position array (32- or 64-bit records) and "listed" bits
    if (head != NONE) {                                                             // else total_anchors == 0 (chain.rs:253)
        const Chunk ck = chunks[slot];
        const uint32_t p = chunk_pair[slot];
        const bool switched = (pairs[p].flags & 4u) != 0;
        mk = inq_mask + (uint64_t)pairs[p].tile0 * (JOIN_TILE / 64);
        s_begin = ck.s_begin; s_end = ck.s_end;
        if constexpr (W::wide) { const CoArr a = wide[p].a_g; ag = a.p; ag64 = a.is64; qoff = wide[p].a_goff[ck.qctg]; }
        else { ag = pairs[p].a_g; qoff = ck.qoff; }
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
    // What is fetched stays RAW -- the position record and the 64-bit word its "listed" bit sits in -- and is put together where it is counted: put
    // together at once (round 2 and 3's first half) every one of the four fetches waited for its own loads, four round trips in a row per chunk
    // instead of none.  Loads are unconditional (an index beyond the chunk reads its last position and counts as nothing): no branch around them.
    constexpr int PF = 4;
    struct Raw { Co v; unsigned long long m; };
    Raw cur[PF] = {}, nxt[PF] = {};
    auto fetch = [&](const Arr& ag_j, const unsigned long long* mk_j, uint32_t s2, uint32_t se_j) -> Raw {
        const uint32_t s = s2 < se_j ? s2 : (se_j ? se_j - 1u : 0u);
        const unsigned long long m = global_of(mk_j)[((s >> 8) << 2) | (s & 3u)];   // join_count_kernel's layout: bit l of word r of a 256-position round = position 4 l + r
        return Raw{(Co)ag_j[s], m};
    };
    auto value = [](const Raw& r, uint32_t s2, uint32_t se_j) -> Co {                // coordinate << 1 | listed; 0 beyond the chunk
        return s2 < se_j ? (r.v & ~(Co)1) | (Co)((r.m >> ((s2 >> 2) & 63u)) & 1ull) : (Co)0;
    };
    auto bcast_arr = [&](int src) -> Arr {                                           // lane src's position array, to all lanes
        const unsigned long long v = (unsigned long long)ag;
        const uint32_t lo32 = wave_readlane((uint32_t)v, src), hi32 = wave_readlane((uint32_t)(v >> 32), src);
        const void* ptr = (const void*)(((unsigned long long)hi32 << 32) | lo32);
        if constexpr (W::wide) return CoArr{ptr, wave_readlane(ag64, src)}; else return global_of((const uint32_t*)ptr);
    };
    auto bcast_ptr = [&](const void* ptr, int src) -> const void* {
        const unsigned long long v = (unsigned long long)ptr;
        const uint32_t lo32 = wave_readlane((uint32_t)v, src), hi32 = wave_readlane((uint32_t)(v >> 32), src);
        return (const void*)(((unsigned long long)hi32 << 32) | lo32);
    };
    int j = -1; uint32_t sb = 0, se = 0;
    Arr agj{}; const unsigned long long* mkj = nullptr;
    if (todo) {
        j = __ffsll((long long)todo) - 1; todo &= todo - 1ull;
        sb = wave_readlane(s_begin, j); se = wave_readlane(s_end, j);
        agj = bcast_arr(j); mkj = (const unsigned long long*)bcast_ptr(mk, j);
#pragma unroll
        for (int u = 0; u < PF; u++) cur[u] = fetch(agj, mkj, sb + 64u * (uint32_t)u + l, se);
    }
    while (j >= 0) {                                                                // wave-uniform
        // the next chunk's first 256 positions are requested now, without a branch around the loads (after the last chunk: the current one once more), so that
        // the wait in front of the counting below is for the CURRENT chunk's words only -- the compiler counts exactly what is in flight behind them
        const bool more = todo != 0;
        const int jn = more ? __ffsll((long long)todo) - 1 : j;
        todo &= todo - 1ull;                                                        // (0 stays 0)
        const uint32_t sbn = wave_readlane(s_begin, jn), sen = wave_readlane(s_end, jn);
        const Arr agn = bcast_arr(jn); const unsigned long long* mkn = (const unsigned long long*)bcast_ptr(mk, jn);
#pragma unroll
        for (int u = 0; u < PF; u++) nxt[u] = fetch(agn, mkn, sbn + 64u * (uint32_t)u + l, sen);
        const uint32_t nj = wave_readlane(n_int, j), q0j = wave_readlane(rq0, j), q1j = wave_readlane(rq1, j), headj = wave_readlane(head, j);
        const Co qoffj = wave_readlane(qoff, j);                                    // positions are padded coordinates; intervals are contig-local
        // the chunk's intervals as (start, width): wave-uniform values, "inside" is one unsigned compare; a chunk has one interval as a rule (mean 1.2),
        // the tests of the others are skipped by scalar branches on their number
        uint32_t lj[STATS_REG], wj[STATS_REG];
#pragma unroll
        for (int i = 0; i < STATS_REG; i++) { lj[i] = 0; wj[i] = 0; if (i == 0 || (uint32_t)i < nj) { lj[i] = wave_readlane(lo[i], j); wj[i] = wave_readlane(hi[i], j) - lj[i]; } }
        const uint32_t qwj = q1j - q0j;
        uint32_t cu = 0, cr = 0, cl = 0;
        auto count = [&](Co v) {
            const bool on = (v & 1u) != 0;                                          // listed in query_positions_all (0 beyond the chunk)
            const uint32_t pos = (uint32_t)((v >> 1) - qoffj);
            bool hit;
            if (nj <= (uint32_t)STATS_REG) {
                hit = pos - lj[0] <= wj[0];                                         // (an active chunk has at least one interval)
                if (nj > 1) { hit = hit || pos - lj[1] <= wj[1]; if (nj > 2) { hit = hit || pos - lj[2] <= wj[2]; if (nj > 3) hit = hit || pos - lj[3] <= wj[3]; } }
            } else {
                hit = false;
                for (uint32_t e = headj; e != NONE; e = ivl_next[e]) { const Interval iv = ivls[e]; const uint32_t l0 = iv.q0 > c ? iv.q0 - c : 0; hit = hit || (pos >= l0 && pos <= iv.q1 + c); }
            }
            cl += (uint32_t)__popcll(__ballot(on));                                 // chain.rs:755-780: seeds of the chunk
            cu += (uint32_t)__popcll(__ballot(on && hit));                          // chain.rs:268-272
            cr += (uint32_t)__popcll(__ballot(on && pos - q0j <= qwj));             // chain.rs:326-332 (spacing estimates are 0)
        };
#pragma unroll
        for (int u = 0; u < PF; u++) if (sb + 64u * (uint32_t)u < se) count(value(cur[u], sb + 64u * (uint32_t)u + l, se));
        for (uint32_t b2 = sb + 64u * PF; b2 < se; b2 += 64) count(value(fetch(agj, mkj, b2 + l, se), b2 + l, se));
        if ((int)l == j) { in_u = cu; in_range = cr; in_list = cl; }
        j = more ? jn : -1; sb = sbn; se = sen; agj = agn; mkj = mkn;
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
