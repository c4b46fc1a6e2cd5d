// chain_join.h -- get_anchors join (chain.rs:608-737): join_count_kernel / join_fill_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ join
// Workgroups are launched in "slots": slot b runs the JOIN_GROUP tiles from slot_tile[b] on (or nothing).  The host interleaves the
// tile groups so that all tiles probing the same sketch B land on the same XCD (block b -> XCD b % 8 on MI355X): B's seed
// table and list storage then stay in that XCD's 4 MiB L2 instead of being fetched by all eight.
// Probes of one tile: R positions per thread.  Everything that can be in flight together is: the R home-slot loads, then the cluster walks in
// lockstep (one more load for every probe that still needs one, as long as any lane of the wave does), then the list heads of the few long lists.
// On return: rec[r] = the slot's payload (TAB_REPETITIVE: no anchors), n_anch[r] = anchors of the position, inq[r] = listed in query_positions_all.
template <int R>
__device__ __forceinline__ void probe_tile(const PairDesc& pd, const uint32_t* bm, bool use_bm, const uint32_t (&h)[R], const bool (&live)[R],
                                           uint32_t (&rec)[R], uint32_t (&n_anch)[R], uint32_t (&inq)[R]) {
    const uint64_t* tab = pd.b_tab;
    uint32_t sl[R]; unsigned long long e[R]; bool more[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        sl[r] = 0; e[r] = TAB_EMPTY;
        if (live[r]) {
            const uint32_t b = seed_bucket(h[r], pd.b_nbk);
            sl[r] = tab_slot(b);
            const uint32_t fb = tab_filter_bits(h[r]);
            if (!use_bm || (bm[b >> TAB_FILTER_SHIFT] & fb) == fb) e[r] = tab[sl[r]];
        }
    }
    // a cluster ascends by hash from the home slot on; TAB_EMPTY (all ones; the last slot of every slice is one) ends every walk.  One 8-byte slot per
    // step: fetching two or four slots per step halves the steps but was slower (2.74 / 2.82 vs 2.33 ms) -- what a gather costs grows with its bytes
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; r++) { more[r] = (uint32_t)(e[r] >> 32) < h[r]; any = any || more[r]; }
    while (__any(any)) {
#pragma unroll
        for (int r = 0; r < R; r++) if (more[r]) e[r] = tab[++sl[r]];
        any = false;
#pragma unroll
        for (int r = 0; r < R; r++) { more[r] = more[r] && (uint32_t)(e[r] >> 32) < h[r]; any = any || more[r]; }
    }
    uint32_t head[R];
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
    for (int r = 0; r < R; r++) if (head[r]) n_anch[r] = pd.b_ms[rec[r] & TAB_OFF_MASK];   // long lists (more than four positions): the count heads the list
}

// A workgroup takes JOIN_GROUP consecutive tiles of one pair, ONE WAVE PER TILE: the bitmap is staged once for the four tiles (the only barrier),
// then every wave walks its tile on its own in four rounds of 256 positions (four probes per lane in flight, the next round's hashes already
// fetched) -- no barrier, no shared scan.  A workgroup's life is a chain of dependent memory round trips; this form needs about half as many per
// tile as "one workgroup = one tile, one round per wave" did (slot, descriptor and staging are shared by four tiles).
// Output per tile: its hits -- the positions with at least one anchor -- as compact (query position, slot payload) records in position order
// (30 % of the positions: the fill pass reads those instead of a probe record plus the position of EVERY position), their number, the number of
// anchors, and one bit per position: "listed in query_positions_all" (chain.rs:682-700), 64 positions per word straight from the ballot.
__global__ __launch_bounds__(256) void join_count_kernel(const PairDesc* pairs, const uint2* slot_tile,
                                                         uint32_t band, uint32_t* tile_anch, uint32_t* tile_hits, uint32_t* pair_anch, uint32_t* pair_inq,
                                                         uint2* hits, unsigned long long* inq_mask, uint32_t lds_words) {
    SKH_DYN_SMEM(smem);
    uint32_t* bm = (uint32_t*)smem;
    const uint2 st = slot_tile[blockIdx.x];                                          // (first tile, pair) in one load
    if (st.x == NONE) return;
    const uint32_t p = st.y;
    const PairDesc pd = pairs[p];
    constexpr int R = 4;
    // B's occupancy filter (common.h: a word per 16 home slots, two bits per seed, ~20 KB) is staged in LDS with coalesced 16-byte loads: 85 % of
    // the probes of absent seeds end there and cost no memory request at all; the others read their home slot -- the entry
    // itself, or the head of the short cluster it sits in (sketch_build.hip build_tables_kernel).  What the gathers cost is not L2 requests or
    // latency but what the L1 can return (profiles/r02_join_count_ablation.md): fewer gathers is what pays.
    const uint32_t bm_words = ((pd.b_nbk + TAB_FILTER_HOMES - 1) / TAB_FILTER_HOMES + 3) / 4 * 4;
    const bool use_bm = bm_words <= lds_words;
    if (use_bm) {
        const uint4* src = (const uint4*)pd.b_bmap;
        for (uint32_t w4 = threadIdx.x; w4 < bm_words / 4; w4 += 256) ((uint4*)bm)[w4] = src[w4];
        __syncthreads();
    }
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const uint32_t tile = st.x + w;                                                  // this wave's tile
    const uint32_t start = (tile - pd.tile0) * JOIN_TILE;
    if (start >= pd.a_n) return;
    uint32_t nh[R], ng[R], nrep[R];
    auto fetch = [&](uint32_t round) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t i = start + round * 256 + r * 64 + l;
            const bool in = i < pd.a_n; const uint32_t gi = pd.a_pos0 + i;
            nh[r] = in ? pd.a_hash[i] : 0u; ng[r] = in ? pd.a_g[i] : 0u;
            nrep[r] = in ? (pd.a_rep[gi >> 5] >> (gi & 31u)) & 1u : 1u;             // chain.rs:674-676: more than `band` positions in A
        }
    };
    fetch(0);
    uint32_t na = 0, nq = 0, n_hit = 0;                                              // n_hit: wave-uniform
    uint2* my_hits = hits + (uint64_t)tile * JOIN_TILE;
    for (uint32_t round = 0; round < JOIN_TILE / 256; round++) {
        if (start + round * 256 >= pd.a_n) break;
        uint32_t h[R], qg[R]; bool live[R];
#pragma unroll
        for (int r = 0; r < R; r++) { h[r] = nh[r]; qg[r] = ng[r]; live[r] = !nrep[r]; }
        if (round + 1 < JOIN_TILE / 256) fetch(round + 1);
        uint32_t rec[R], n_anch[R], inq[R];
        probe_tile<R>(pd, bm, use_bm, h, live, rec, n_anch, inq);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t o = round * 256 + r * 64 + l;
            const unsigned long long m = __ballot(inq[r] != 0);
            if (l == 0) inq_mask[(uint64_t)tile * (JOIN_TILE / 64) + (o >> 6)] = m;
            // hit record = (query position, the slot's payload: B's position itself or the reference to the seed's position list)
            const unsigned long long hm = __ballot(n_anch[r] != 0);
            if (n_anch[r]) my_hits[n_hit + (uint32_t)__popcll(hm & ((1ull << l) - 1ull))] = make_uint2(qg[r], rec[r]);
            n_hit += (uint32_t)__popcll(hm);
            na += n_anch[r]; nq += inq[r];
        }
    }
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
__global__ __launch_bounds__(256) void join_fill_kernel(const PairDesc* pairs, const uint2* slot_tile, uint32_t tile_base, const uint32_t* toff_a,
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
            const uint32_t* bs = pd.b_ms + (rec[r] & TAB_OFF_MASK);
            n_anch[r] = rec[r] == TAB_REPETITIVE ? 0u : (!listed ? 1u : (code ? code + 1u : bs[0]));
            f0[r] = listed ? bs[1] : rec[r]; f1[r] = listed ? bs[2] : 0u;             // (behind a list of one -- a single beyond 2^31 -- follows another list or the storage's slack)
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t incl = wave_incl_scan(n_anch[r]);
            const uint32_t tot = wave_readlane(incl, 63);
            if (n_anch[r]) {
                uint32_t oa = run + incl - n_anch[r];
                const uint32_t* bs = pd.b_ms + (rec[r] & TAB_OFF_MASK) + 1;
                for (uint32_t k = 0; k < n_anch[r]; k++, oa++) {                     // chain.rs:703-711, already in sorted order
                    const uint32_t rg = k == 0 ? f0[r] : (k == 1 ? f1[r] : bs[k]);
                    anc_q[oa] = qg[r] >> 1; anc_r[oa] = (rg & ~1u) | ((rg ^ qg[r]) & 1u);
                }
            }
            run += tot;
        }
    }
}
