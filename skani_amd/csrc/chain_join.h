// chain_join.h -- get_anchors join (chain.rs:608-737): join_count_kernel / join_fill_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ join
// Workgroups are launched in "slots": slot b runs the JOIN_GROUP tiles from slot_tile[b] on (or nothing).  The host interleaves the
// tile groups so that all tiles probing the same sketch B land on the same XCD (block b -> XCD b % 8 on MI355X): B's seed
// table and list storage then stay in that XCD's 4 MiB L2 instead of being fetched by all eight.
// The count pass.  A workgroup takes JOIN_GROUP consecutive tiles of one pair, ONE WAVE PER TILE: B's occupancy filter is staged once for the four tiles
// (the only barrier), then every wave walks its tile on its own in four rounds of 256 positions, the next round's hashes already fetched -- no barrier,
// no shared scan.  Output per tile: its hits -- the positions with at least one anchor -- as compact (query position, slot payload) records in position
// order (30 % of the positions: the fill pass reads those instead of a probe record plus the position of EVERY position), their number, the number of
// anchors, and one bit per position: "listed in query_positions_all" (chain.rs:682-700).
// Round 3 rebuilt the round around the number of vector-memory instructions it issues (52 per round before, ~18 now; 2.33 -> 2.22 ms; what else was
// tried -- tiles of A shared by its partners, a denser filter, more probes in flight -- is in profiles/r03_join_variants.md):
//   * a lane owns FOUR CONSECUTIVE positions of the round: their hashes and positions are two 16-byte loads (was eight 4-byte loads), the four
//     'repetitive' bits come out of nine words fetched by nine lanes (was four loads);
//   * the probes that pass the occupancy filter (45 %) are COMPACTED through a small LDS queue and worked off two per lane: a home-slot read and the
//     cluster-walk steps are issued for dense lanes only -- a walk step was up to four nearly empty instructions per iteration of the lockstep loop;
//   * the round's four "listed" words are one store (bit l of word r = position 4 l + r of the round: chunk_stats_kernel reads that layout), the hit
//     records go out per lane in position order from one prefix over the wave.
struct __attribute__((packed, aligned(4))) Words4 { uint32_t x, y, z, w; };   // four consecutive words at a 4-byte aligned address
struct __attribute__((packed, aligned(8))) Slots2 { unsigned long long a, b; };   // two consecutive table slots
constexpr uint32_t JOIN_Q = 128;     // dense probe queue of a wave: 8 bytes per entry; a round has 115 +- 8 passing probes, more take a second turn
// exclusive prefix / total over the wave of a per-lane count c <= 7: three ballots instead of a six-step scan
__device__ __forceinline__ uint32_t wave_excl_small(uint32_t c, uint32_t l, uint32_t& total) {
    const unsigned long long b0 = __ballot((c & 1u) != 0), b1 = __ballot((c & 2u) != 0), b2 = __ballot((c & 4u) != 0), lt = (1ull << l) - 1ull;
    total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
    return (uint32_t)__popcll(b0 & lt) + 2u * (uint32_t)__popcll(b1 & lt) + 4u * (uint32_t)__popcll(b2 & lt);
}
#ifdef SKH_JOIN_WAVES   // experiment builds (tools/exp/build_variants.py chain.hip SKH_JOIN_WAVES 6 8): the register allocator held to that many waves per SIMD
#define SKH_JOIN_OCC __attribute__((amdgpu_waves_per_eu(SKH_JOIN_WAVES, SKH_JOIN_WAVES)))
#else
#define SKH_JOIN_OCC
#endif
template <bool PROF>
__global__ __launch_bounds__(JOIN_THREADS) SKH_JOIN_OCC void join_count_kernel(const PairDesc* pairs, const uint2* slot_tile,
                                                         uint32_t band, uint32_t* tile_anch, uint32_t* tile_hits, uint32_t* pair_anch, uint32_t* pair_inq,
                                                         uint2* hits, unsigned long long* inq_mask, uint32_t lds_words, unsigned long long* prof = nullptr) {
    SKH_DYN_SMEM(smem);
    uint32_t* bm = (uint32_t*)smem;
    const uint2 st = slot_tile[blockIdx.x];                                          // (first tile, pair) in one load
    if (st.x == NONE) return;
    const uint32_t p = st.y;
    const PairDesc pd = pairs[p];
    constexpr int R = 4;
    constexpr uint32_t RW = 64u * R;                                                 // positions per round
    const uint32_t bm_words = ((pd.b_nbk + TAB_FILTER_HOMES - 1) / TAB_FILTER_HOMES + 3) / 4 * 4;
    const bool use_bm = bm_words <= lds_words;
    // The sketches' arrays are reached through pointers that were loaded from the pair record: generic pointers to the compiler, whose loads (flat_load)
    // count on the LDS wait counter as well -- every wait for the filter or the probe queue then also waits for the hashes requested a round ahead.
    // global_of() (dev.h) names the address space; the loads become global_load and the waits exact.
    const GlobalPtr<uint32_t> a_seed = global_of(pd.a_seed), a_g = global_of(pd.a_g), a_rep = global_of(pd.a_rep), b_ms = global_of(pd.b_ms);
    if (use_bm) {                                                                    // B's occupancy filter, staged once for the workgroup's tiles (the only barrier)
        const GlobalPtr<uint4> src = (GlobalPtr<uint4>)global_of(pd.b_bmap);
        for (uint32_t w4 = threadIdx.x; w4 < bm_words / 4; w4 += JOIN_THREADS) ((uint4*)bm)[w4] = src[w4];
        __syncthreads();
    }
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    unsigned long long* q = (unsigned long long*)(bm + lds_words) + (size_t)w * JOIN_Q;   // this wave's probe queue
    const uint32_t tile = st.x + w;                                                  // this wave's tile
    const uint32_t start = (tile - pd.tile0) * JOIN_TILE;
    if (start >= pd.a_n) return;
    const GlobalPtr<uint64_t> tab = global_of(pd.b_tab);
    // position start + round * 256 + 4 l + r sits in slot r of lane l.  A genome's arrays start at any element of the set's arrays: the four-word loads
    // are only 4-byte aligned (Words4), which global_load_dwordx4 takes.
    uint32_t nh[R], ng[R], nrep;
    auto fetch = [&](uint32_t round) {
        const uint32_t i0 = start + round * RW + 4u * l;
        if (i0 + 4u <= pd.a_n) {
            const Words4 a = *(GlobalPtr<Words4>)(a_seed + i0), b = *(GlobalPtr<Words4>)(a_g + i0);
            nh[0] = a.x; nh[1] = a.y; nh[2] = a.z; nh[3] = a.w; ng[0] = b.x; ng[1] = b.y; ng[2] = b.z; ng[3] = b.w;
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) { const bool in = i0 + (uint32_t)r < pd.a_n; nh[r] = in ? a_seed[i0 + r] : 0u; ng[r] = in ? a_g[i0 + r] : 0u; }
        }
        // 'repetitive' bits (chain.rs:674-676: more than `band` positions in A) of the round's 256 positions: bits [x0, x0 + 256) of the words from rw0 on
        const uint32_t gi0 = pd.a_pos0 + start + round * RW, rw0 = gi0 >> 5, x = (gi0 & 31u) + 4u * l;
        const uint32_t last_w = (pd.a_pos0 + pd.a_n - 1u) >> 5;
        const uint32_t word = (l < 9u && rw0 + l <= last_w) ? a_rep[rw0 + l] : 0u;
        const uint32_t lo = __shfl(word, (int)(x >> 5), 64), hi = __shfl(word, (int)((x >> 5) + 1u), 64);
        nrep = (uint32_t)(((((unsigned long long)hi << 32) | lo) >> (x & 31u)) & 0xFull);
#pragma unroll
        for (int r = 0; r < R; r++) if (i0 + (uint32_t)r >= pd.a_n) nrep |= 1u << r;   // beyond the sketch: nothing to probe
    };
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;                 // PROF: cycles per phase of this wave, walk iterations in tk[7]
    auto tick = [&](int ph) { if (PROF) { const unsigned long long t = wave_clock(); tk[ph] += t - t_prev; t_prev = t; } };
    if (PROF) t_prev = wave_clock();
    fetch(0);
    uint32_t na = 0, nq = 0, n_hit = 0;                                              // n_hit: wave-uniform
    uint2* my_hits = hits + (uint64_t)tile * JOIN_TILE;
    for (uint32_t round = 0; round < JOIN_TILE / RW; round++) {
        if (start + round * RW >= pd.a_n) break;
        uint32_t h[R], qg[R], sl[R]; bool live[R], pass[R];
        const uint32_t rep = nrep;
#pragma unroll
        for (int r = 0; r < R; r++) { h[r] = table_hash(nh[r], pd.b_salt); qg[r] = ng[r]; live[r] = !((rep >> r) & 1u); }   // (the seeds are stored, not their hashes: 4 bytes less per position, and B decides the salt)
        if (PROF) { wait_for_value(h[0]); wait_for_value(qg[3]); tick(0); }                 // 0: the round's hashes / positions have arrived
        if (round + 1 < JOIN_TILE / RW) fetch(round + 1);
        // the occupancy filter (common.h): 85 % of the probes of absent seeds end here
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t b = seed_bucket(h[r], pd.b_nbk), fb = tab_filter_bits(h[r]);
            sl[r] = tab_slot(b);
            pass[r] = live[r] && (!use_bm || (bm[b >> TAB_FILTER_SHIFT] & fb) == fb);
            c += pass[r] ? 1u : 0u;
        }
        uint32_t total;
        const uint32_t excl = wave_excl_small(c, l, total);
        tick(1);                                                                     // 1: next fetch issued, filter evaluated, prefix
        // the passing probes, dense: (hash, slot) into the queue, two per lane walk their clusters in lockstep, the slot found comes back through the queue.
        // A cluster ascends by hash from the home slot on; TAB_EMPTY (all ones; the last slot of every slice is one) ends every walk.
        unsigned long long e[R];
#pragma unroll
        for (int r = 0; r < R; r++) e[r] = TAB_EMPTY;
        for (uint32_t base = 0; base < total; base += JOIN_Q) {
            uint32_t d = excl - base;                                                // (wraps for probes of an earlier turn: they fail the bound test below)
#pragma unroll
            for (int r = 0; r < R; r++) if (pass[r]) { if (d < JOIN_Q) q[d] = ((unsigned long long)h[r] << 32) | sl[r]; d++; }
            wave_sync_mem();
            const uint32_t n_here = total - base < JOIN_Q ? total - base : JOIN_Q;
            const bool v0 = l < n_here, v1 = 64u + l < n_here;
            const unsigned long long x0 = v0 ? q[l] : ~0ull, x1 = v1 ? q[64u + l] : ~0ull;
            const uint32_t ph0 = (uint32_t)(x0 >> 32), ph1 = (uint32_t)(x1 >> 32);
            uint32_t ps0 = (uint32_t)x0, ps1 = (uint32_t)x1;
            tick(2);                                                                 // 2: queue written and read
            // a walk reads TWO consecutive slots per request (16 bytes at an 8-byte aligned address): the lockstep loop below is a chain of dependent
            // round trips, as many as the longest walk of the round has steps -- pairs of slots halve them: 2.25 -> 2.15 ms (four slots per request:
            // no further gain, 2.13-2.15 ms).  Behind a slice's last slot, which is empty and ends every walk, lies the next slice or the table's
            // slack: readable, never used.
            auto ld = [&](uint32_t ps) { return *(GlobalPtr<Slots2>)(tab + ps); };
            const Slots2 none{TAB_EMPTY, TAB_EMPTY};
            Slots2 s0 = v0 ? ld(ps0) : none, s1 = v1 ? ld(ps1) : none;
            auto settle = [](const Slots2& s, uint32_t ph, unsigned long long& e) {       // the first slot whose hash is not below the probe's, if it is here
                if ((uint32_t)(s.a >> 32) >= ph) { e = s.a; return false; }
                if ((uint32_t)(s.b >> 32) >= ph) { e = s.b; return false; }
                return true;
            };
            if (PROF) { wait_for_value((uint32_t)s0.a); wait_for_value((uint32_t)s1.a); tick(3); }   // 3: home slots have arrived
            unsigned long long e0 = TAB_EMPTY, e1 = TAB_EMPTY;
            bool m0 = settle(s0, ph0, e0), m1 = settle(s1, ph1, e1);
            while (__any(m0 || m1)) {
                if (PROF) tk[7]++;
                if (m0) { ps0 += 2; s0 = ld(ps0); }
                if (m1) { ps1 += 2; s1 = ld(ps1); }
                m0 = m0 && settle(s0, ph0, e0); m1 = m1 && settle(s1, ph1, e1);
            }
            tick(4);                                                                 // 4: cluster walks
            if (v0) q[l] = e0;
            if (v1) q[64u + l] = e1;
            wave_sync_mem();
            d = excl - base;
#pragma unroll
            for (int r = 0; r < R; r++) if (pass[r]) { if (d < JOIN_Q) e[r] = q[d]; d++; }
            wave_sync_mem();
        }
        tick(5);                                                                     // 5: results back through the queue
        uint32_t rec[R], n_anch[R], inq[R], head[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            rec[r] = TAB_REPETITIVE; n_anch[r] = 0; inq[r] = 0; head[r] = 0;
            if (!live[r]) continue;
            const unsigned long long x = e[r];
            if (x == TAB_EMPTY || (uint32_t)(x >> 32) != h[r]) { inq[r] = 1; continue; }   // absent in B: chain.rs:682-685
            const uint32_t xl = (uint32_t)x;
            if (xl == TAB_REPETITIVE) continue;                                            // chain.rs:694-696: dropped entirely
            inq[r] = 1; rec[r] = xl;
            const uint32_t code = tab_list_code(xl);
            if (!(xl & TAB_LISTED)) n_anch[r] = 1u; else if (code) n_anch[r] = code + 1u; else head[r] = 1;
        }
#pragma unroll
        for (int r = 0; r < R; r++) if (head[r]) n_anch[r] = b_ms[rec[r] & TAB_OFF_MASK];   // long lists (more than four positions): the count heads the list
        // "listed in query_positions_all" (chain.rs:682-700): four words per round, bit l of word r = position 4 l + r of the round
        unsigned long long mine = 0;
#pragma unroll
        for (int r = 0; r < R; r++) { const unsigned long long m = __ballot(inq[r] != 0); if (l == (uint32_t)r) mine = m; }
        if (l < (uint32_t)R) inq_mask[(uint64_t)tile * (JOIN_TILE / 64) + round * R + l] = mine;
        // hit records = (query position, the slot's payload: B's position itself or the reference to the seed's position list), in position order
        uint32_t hc = 0, n_round;
#pragma unroll
        for (int r = 0; r < R; r++) hc += n_anch[r] ? 1u : 0u;
        uint32_t o = n_hit + wave_excl_small(hc, l, n_round);
#pragma unroll
        for (int r = 0; r < R; r++) { if (n_anch[r]) my_hits[o++] = make_uint2(qg[r], rec[r]); na += n_anch[r]; nq += inq[r]; }
        n_hit += n_round;
        tick(6);                                                                     // 6: classification, list heads, "listed" words, hit records
    }
    if (PROF && l == 0) { for (int x = 0; x < 8; x++) atomicAdd(&prof[x], tk[x]); atomicAdd(&prof[8], 1ull); }
    na = wave_sum(na); nq = wave_sum(nq);
    if (l == 0) {
        tile_anch[tile] = na; tile_hits[tile] = n_hit;
        if (na) atomicAdd(&pair_anch[p], na);
        if (nq) atomicAdd(&pair_inq[p], nq);
    }
}

// Emits the anchors of the tiles at the offsets given by the tile scan, from the hit records of join_count_kernel (no second probe).  Same
// shape: four tiles per workgroup, a wave per tile, 256 hits per round with the wave's own running offset -- no barrier at all.  The pass moves
// 0.9 GB in and 2.4 GB out in 0.50 ms (6.6 TB/s): HBM-bound; two tiles per wave with the second tile's records prefetched changed nothing (0.52 ms).
__global__ __launch_bounds__(JOIN_THREADS) void join_fill_kernel(const PairDesc* pairs, const uint2* slot_tile, uint32_t tile_base, const uint32_t* toff_a,
                                                        const uint32_t* tile_hits, const uint2* hits, uint32_t* anc_q, uint32_t* anc_r) {
    constexpr int R = 4;
    const uint2 st = slot_tile[blockIdx.x];
    if (st.x == NONE) return;
    const PairDesc pd = pairs[st.y];
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const uint32_t tile = st.x + w, start = (tile - pd.tile0) * JOIN_TILE;
    if (start >= pd.a_n) return;
    const uint32_t n_hit = tile_hits[tile];
    const uint2* my_hits = hits + (uint64_t)tile * JOIN_TILE;
    uint32_t run = toff_a[tile - tile_base];
    uint2 nx[R];
    auto fetch = [&](uint32_t x0) {
#pragma unroll
        for (int r = 0; r < R; r++) { const uint32_t x = x0 + r * 64 + l; nx[r] = x < n_hit ? my_hits[x] : make_uint2(0u, TAB_REPETITIVE); }
    };
    fetch(0);
    for (uint32_t x0 = 0; x0 < n_hit; x0 += 256) {
        uint32_t rec[R], qg[R], n_anch[R], f0[R], f1[R];
#pragma unroll
        for (int r = 0; r < R; r++) { qg[r] = nx[r].x; rec[r] = nx[r].y; }
        if (x0 + 256 < n_hit) fetch(x0 + 256);
        // counts from the payload; long lists keep theirs at the list head.  The first two positions of every list are fetched for all R records at
        // once (most lists have two), together with those heads.
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool listed = rec[r] != TAB_REPETITIVE && (rec[r] & TAB_LISTED);
            const uint32_t code = tab_list_code(rec[r]);
            const GlobalPtr<uint32_t> bs = global_of(pd.b_ms) + (rec[r] & TAB_OFF_MASK);
            n_anch[r] = rec[r] == TAB_REPETITIVE ? 0u : (!listed ? 1u : (code ? code + 1u : bs[0]));
            f0[r] = listed ? bs[1] : rec[r]; f1[r] = listed ? bs[2] : 0u;             // (behind a list of one -- a single beyond 2^31 -- follows another list or the storage's slack)
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t incl = wave_incl_scan(n_anch[r]);
            const uint32_t tot = wave_readlane(incl, 63);
            if (n_anch[r]) {
                uint32_t oa = run + incl - n_anch[r];
                const GlobalPtr<uint32_t> bs = global_of(pd.b_ms) + (rec[r] & TAB_OFF_MASK) + 1;
                for (uint32_t k = 0; k < n_anch[r]; k++, oa++) {                     // chain.rs:703-711, already in sorted order
                    const uint32_t rg = k == 0 ? f0[r] : (k == 1 ? f1[r] : bs[k]);
                    anc_q[oa] = qg[r] >> 1; anc_r[oa] = (rg & ~1u) | ((rg ^ qg[r]) & 1u);
                }
            }
            run += tot;
        }
    }
}
