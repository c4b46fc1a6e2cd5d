// chain_join.h -- get_anchors join (chain.rs:608-737): join_count_kernel / join_fill_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ join
// Workgroups are launched in "slots": slot b runs logical tile slot_tile[b] (or nothing).  The host interleaves the
// tiles so that all tiles probing the same sketch B land on the same XCD (block b -> XCD b % 8 on MI355X): B's hash
// table and seed-order arrays then stay in that XCD's 4 MiB L2 instead of being fetched by all eight.
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
            if (!use_bm || ((bm[b >> 5] >> (b & 31u)) & 1u)) e[r] = tab[sl[r]];
        }
    }
    // a cluster ascends by hash from the home slot on; TAB_EMPTY (all ones; the slack slots behind every slice end with one) ends every walk
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
    constexpr int R = JOIN_TILE / 256;
    // B's bucket-occupancy bitmap (1 bit per home slot of its seed table, ~10 KB) is staged in LDS with coalesced 16-byte loads: 61 % of
    // the buckets are nobody's home, and a probe of one costs no memory request at all; the others read their home slot -- the entry
    // itself, or the head of the short cluster it sits in (sketch_build.hip build_tables_kernel).  The kernel runs at the L2's
    // request rate (one 64-byte line per random 8-byte read), so requests are what to save.
    const uint32_t bm_words = ((pd.b_nbk + 31) / 32 + 3) / 4 * 4;
    const bool use_bm = bm_words <= lds_words;
    if (use_bm) {
        const uint4* src = (const uint4*)pd.b_bmap;
        for (uint32_t w4 = threadIdx.x; w4 < bm_words / 4; w4 += 256) ((uint4*)bm)[w4] = src[w4];
        __syncthreads();
    }
    // the four positions of this thread are probed together: their loads are independent, so they overlap
    uint32_t h[R]; bool live[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t i = start + r * 256 + threadIdx.x;
        live[r] = i < pd.a_n;
        const uint32_t gi = pd.a_pos0 + i;
        const uint32_t rep = live[r] ? (pd.a_rep[gi >> 5] >> (gi & 31u)) & 1u : 1u;
        h[r] = live[r] ? pd.a_hash[i] : 0u;
        live[r] = live[r] && !rep;                                                 // chain.rs:674-676: more than `band` positions in A
    }
    uint32_t rec[R], n_anch[R], inq[R];
    probe_tile<R>(pd, bm, use_bm, h, live, rec, n_anch, inq);
    uint32_t na = 0, nq = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t o = r * 256 + threadIdx.x, i = start + o;
        // probe record = the slot's payload (B's position itself, or the reference to the seed's position list; TAB_REPETITIVE = no anchors); and
        // one bit per position: "listed in query_positions_all" (chain.rs:682-700), 64 positions per word straight from the ballot
        if (i < pd.a_n) pinfo[(uint64_t)tile * JOIN_TILE + o] = rec[r];
        const unsigned long long m = __ballot(inq[r] != 0);
        if ((threadIdx.x & 63) == 0) inq_mask[(uint64_t)tile * (JOIN_TILE / 64) + (o >> 6)] = m;
        na += n_anch[r]; nq += inq[r];
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
    uint32_t n_anch[R], qg[R], rec[R], ia[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t o = r * 256 + threadIdx.x, i = start + o;
        rec[r] = TAB_REPETITIVE; qg[r] = 0;
        if (i < pd.a_n) { rec[r] = pinfo[(uint64_t)tile * JOIN_TILE + o]; qg[r] = pd.a_g[i]; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t code = tab_list_code(rec[r]);
        n_anch[r] = rec[r] == TAB_REPETITIVE ? 0u : (!(rec[r] & TAB_LISTED) ? 1u : (code ? code + 1u : pd.b_ms[rec[r] & TAB_OFF_MASK]));
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
            uint32_t oa = run_a + ba + ia[r] - n_anch[r];
            if (!(rec[r] & TAB_LISTED)) { anc_q[oa] = qg[r] >> 1; anc_r[oa] = (rec[r] & ~1u) | ((rec[r] ^ qg[r]) & 1u); }
            else {
                const uint32_t* bs = pd.b_ms + (rec[r] & TAB_OFF_MASK) + 1;
                for (uint32_t k = 0; k < n_anch[r]; k++, oa++) {                     // chain.rs:703-711, already in sorted order
                    const uint32_t rg = bs[k];
                    anc_q[oa] = qg[r] >> 1; anc_r[oa] = (rg & ~1u) | ((rg ^ qg[r]) & 1u);
                }
            }
        }
        run_a += ta;
    }
}

