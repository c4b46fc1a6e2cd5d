// chain_dp.h -- chain_anchors_ani + get_chain_intervals (chain.rs:838-1007): chain_dp_thread_kernel, chain_dp_kernel, interval_emit_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ banded chaining DP
// chain.rs:838-896 + score_anchors :558-603.  One wave per chunk.  Lanes own anchors base..base+63; sources j are
// swept in increasing order; a source's score is final when the sweep reaches it, so it is broadcast with v_readlane.
// All values are integers (positions, 20, gap) => int32 is exact where the reference uses f64.
template <class Co> struct Blk { Co q, r; uint32_t cr; int32_t score; uint32_t root, depth; };

// PB = number of earlier 64-anchor blocks kept in registers (band <= 64 PB).  PB = 0: any band (c < 10: up to 2500 anchors) -- the state of earlier
// anchors (score, root, depth) goes through a per-anchor record array in memory instead; read past the L1, the same wave wrote it a block ago.
// W (chain_types.h): 32- or 64-bit coordinates; a Wide run chains with this kernel whatever its band.
template <int PB, class W>
__global__ __launch_bounds__(256) void chain_dp_kernel(uint32_t n_slots, const Chunk* chunks, uint32_t band, const typename W::Co* anc_q, const typename W::Co* anc_r, unsigned long long* best,
                                                       uint32_t* st /* PB == 0: 3 words per anchor */) {
    using Co = typename W::Co; using Blk = skh::Blk<Co>;
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slot >= n_slots) return;
    const Chunk ck = chunks[slot];
    if (ck.a_end <= ck.a_begin) return;
    const int l = (int)lane_id();
    Blk prev[PB > 0 ? PB : 1];
#pragma unroll
    for (int b = 0; b < (PB > 0 ? PB : 1); b++) prev[b] = Blk{0, 0, 0, 0, 0, 0};
    for (uint32_t base = ck.a_begin; base < ck.a_end; base += 64) {
        const uint32_t t = base + (uint32_t)l;
        const bool valid = t < ck.a_end;
        Blk cur;
        Co av_q = 0, av_r = 0;
        if (valid) { av_q = anc_q[t]; av_r = anc_r[t]; }
        cur.q = av_q; cur.r = av_r >> 1; cur.cr = (uint32_t)(av_r & 1u);               // cr: strand only -- different contigs are > MAX_LIN apart
        cur.score = 0; cur.root = t; cur.depth = 1;
        uint32_t ptr = t;
        uint32_t jlo = base - ck.a_begin > band ? base - band : ck.a_begin;
        const uint32_t jhi = ck.a_end < base + 64 ? ck.a_end : base + 64;
        {   // anchors ascend in q: sources more than BP_CHAIN_BAND below this block's first target cannot link to any of its targets
            const Co q_base = anc_q[base];
            uint32_t lo = jlo, hi = base;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (anc_q[mid] + BP_CHAIN_BAND < q_base) lo = mid + 1; else hi = mid; }
            jlo = lo;
        }
        for (uint32_t j = jlo; j < jhi; j++) {
            Co qj, rj; uint32_t crj; int32_t sj;
            if (j >= base) {
                const int ln = (int)(j - base);
                // finalise lane ln: its score/ptr can no longer change (all its predecessors were swept)
                const uint32_t pj = wave_readlane(ptr, ln);
                uint32_t rootj = j, depthj = 1;
                if (pj != j) {
                    if (pj >= base) { rootj = wave_readlane(cur.root, (int)(pj - base)); depthj = wave_readlane(cur.depth, (int)(pj - base)) + 1; }
                    else if (PB == 0) { rootj = __atomic_load_n(&st[3 * (size_t)pj + 1], __ATOMIC_RELAXED); depthj = __atomic_load_n(&st[3 * (size_t)pj + 2], __ATOMIC_RELAXED) + 1; }
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
            } else if (PB == 0) {
                const Co rr = anc_r[j];
                qj = anc_q[j]; rj = rr >> 1; crj = (uint32_t)(rr & 1u); sj = (int32_t)__atomic_load_n(&st[3 * (size_t)j], __ATOMIC_RELAXED);
            } else {
                qj = rj = 0; crj = 0; sj = 0;
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
                const Co dq = cur.q - qj;
                const bool rev = (crj & 1u) != 0;
                const bool fwd_ok = rev ? (rj > cur.r) : (cur.r > rj);
                const Co dr = rev ? rj - cur.r : cur.r - rj;
                if (dq != 0 && dq <= BP_CHAIN_BAND && fwd_ok && dr <= (Co)MAX_LIN) {
                    const int32_t gap = (int32_t)dr > (int32_t)dq ? (int32_t)(dr - dq) : (int32_t)(dq - dr);
                    const int32_t s = ANCHOR_SCORE - gap + sj;
                    // reference scans j downwards and replaces only on strictly greater => among equal maxima the largest j wins
                    if (gap <= MAX_GAP && s > 0 && s >= cur.score) { cur.score = s; ptr = j; }
                }
            }
        }
        if (valid) atomicMax(&best[cur.root], best_payload((uint32_t)cur.score, t - ck.a_begin, cur.depth));   // chain.rs:952-964
        if (PB == 0) {
            if (valid) { __atomic_store_n(&st[3 * (size_t)t], (uint32_t)cur.score, __ATOMIC_RELAXED); __atomic_store_n(&st[3 * (size_t)t + 1], cur.root, __ATOMIC_RELAXED); __atomic_store_n(&st[3 * (size_t)t + 2], cur.depth, __ATOMIC_RELAXED); }
            wave_sync_mem();
        }
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
template <class W> struct EmitCtxT { const typename W::Co *anc_q, *anc_r; const PairDesc* pairs; const WidePair* wide; const uint32_t *pc0, *pi0; uint32_t* ivl_cnt; Interval* ivls; uint32_t* err; };
using EmitCtx = EmitCtxT<Narrow>;
__device__ __forceinline__ bool dp_keep(unsigned long long b) {                  // chain.rs:954-957, 974-977
    const uint32_t sc = (uint32_t)(b >> 40), na = (uint32_t)(b & 0xFFFFFu);
    return na >= MIN_ANCHORS && (int32_t)sc >= MIN_SCORE;
}
// writes the record of a kept chain as the pair's k-th candidate interval
template <class W>
__device__ __forceinline__ void dp_write(const Chunk& ck, uint32_t slot, uint32_t p, uint32_t root, unsigned long long b, const EmitCtxT<W>& ec, uint32_t k) {
    using Co = typename W::Co;
    const uint32_t sc = (uint32_t)(b >> 40), bi = (uint32_t)((b >> 20) & 0xFFFFFu), na = (uint32_t)(b & 0xFFFFFu);
    if (ec.pi0[p] + k >= ec.pi0[p + 1]) { atomicAdd(ec.err, 1u); return; }
    const Co ar_q = ec.anc_q[ck.a_begin + root], ar_r = ec.anc_r[ck.a_begin + root], ab_q = ec.anc_q[ck.a_begin + bi], ab_r = ec.anc_r[ck.a_begin + bi];
    const PairDesc& pd = ec.pairs[p];
    const typename W::Arr bo = W::b_goff(pd, ec.wide, p);
    const Co ra = ar_r >> 1, rb = ab_r >> 1;
    const uint32_t rctg = ctg_of(bo, pd.b_nctg, ra); const Co roff = bo[rctg];
    Co qoff = ck.qoff;                                                               // (a Wide run's chunks do not carry their contig's start)
    if constexpr (W::wide) qoff = W::a_goff(pd, ec.wide, p)[ck.qctg];
    Interval iv;
    iv.score = sc; iv.na = na; iv.q0 = (uint32_t)(ar_q - qoff); iv.q1 = (uint32_t)(ab_q - qoff);
    iv.r0 = (uint32_t)((ra < rb ? ra : rb) - roff); iv.r1 = (uint32_t)((ra < rb ? rb : ra) - roff);
    iv.rctg = rctg; iv.qctg = ck.qctg; iv.chunk = slot - ec.pc0[p]; iv.rev = (uint32_t)(ar_r & 1u);
    ec.ivls[ec.pi0[p] + k] = iv;
}
template <class W>
__device__ __forceinline__ void dp_emit(const Chunk& ck, uint32_t slot, uint32_t p, uint32_t root, unsigned long long b, const EmitCtxT<W>& ec) {
    if (dp_keep(b)) dp_write(ck, slot, p, root, b, ec, atomicAdd(&ec.ivl_cnt[p], 1u));
}

// The 64 lanes of a wave step through their chunks in lockstep, so a wave takes as long as its longest chunk: chunks are
// handed out in order of decreasing anchor count, which puts chunks of nearly equal length side by side (in slot order a wave's lanes are busy only
// ~1/3 of the time: mean 131 anchors, longest of 64 ~350).  The order is a counting sort on the 1024 chunk lengths in two small kernels (a histogram
// and a scatter; which of two equally long chunks comes first is left to the atomics -- it only decides which lane works on which).  Round 2 used
// rocPRIM's radix sort for this: eight launches, 130 us in front of the DP.
constexpr uint32_t DP_ORDER_KEYS = 1024, DP_ORDER_BLOCK = 8192;                      // one class per chunk length (the last one: 1023 and more); chunks per workgroup
__device__ __forceinline__ uint32_t dp_order_key(const Chunk& c) { const uint32_t len = c.a_end - c.a_begin; return (DP_ORDER_KEYS - 1u) - (len >= DP_ORDER_KEYS - 1u ? DP_ORDER_KEYS - 1u : len); }
__global__ __launch_bounds__(256) void dp_order_hist_kernel(uint32_t n_slots, const Chunk* chunks, uint32_t* hist) {
    __shared__ uint32_t lh[DP_ORDER_KEYS];
    for (uint32_t x = threadIdx.x; x < DP_ORDER_KEYS; x += 256) lh[x] = 0;
    __syncthreads();
    const uint32_t e = (blockIdx.x + 1u) * DP_ORDER_BLOCK < n_slots ? (blockIdx.x + 1u) * DP_ORDER_BLOCK : n_slots;
    for (uint32_t i = blockIdx.x * DP_ORDER_BLOCK + threadIdx.x; i < e; i += 256) atomicAdd(&lh[dp_order_key(chunks[i])], 1u);
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < DP_ORDER_KEYS; x += 256) if (lh[x]) atomicAdd(&hist[x], lh[x]);
}
__global__ __launch_bounds__(256) void dp_order_scatter_kernel(uint32_t n_slots, const Chunk* chunks, const uint32_t* hist, uint32_t* cursor, uint32_t* order) {
    __shared__ uint32_t base[DP_ORDER_KEYS], mine[DP_ORDER_KEYS]; __shared__ uint32_t wsum[4];
    // exclusive prefix of the histogram, redone by every workgroup (1024 values): thread t owns classes 4t .. 4t + 3
    const uint32_t t = threadIdx.x;
    uint32_t h[4], s = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) { h[x] = hist[4 * t + x]; s += h[x]; mine[4 * t + x] = 0; }
    const uint32_t incl = wave_incl_scan(s);
    if ((t & 63u) == 63u) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t off = incl - s;
    for (uint32_t w = 0; w < (t >> 6); w++) off += wsum[w];
#pragma unroll
    for (int x = 0; x < 4; x++) { base[4 * t + x] = off; off += h[x]; }
    __syncthreads();
    // the workgroup's chunks: rank within the workgroup by an LDS counter per class, then ONE global reservation per class the workgroup holds
    // (the unused chunk slots of all pairs share the last class: a global atomic per chunk queued a quarter of a million of them on one address: 3.8 ms)
    constexpr int PER = DP_ORDER_BLOCK / 256;
    const uint32_t i0 = blockIdx.x * DP_ORDER_BLOCK + t;
    uint16_t k[PER], r[PER];
#pragma unroll
    for (int x = 0; x < PER; x++) { const uint32_t i = i0 + 256u * x; k[x] = 0xFFFF; if (i < n_slots) { k[x] = (uint16_t)dp_order_key(chunks[i]); r[x] = (uint16_t)atomicAdd(&mine[k[x]], 1u); } }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 4; x++) { const uint32_t c = 4 * t + x, n = mine[c]; if (n) base[c] += atomicAdd(&cursor[c], n); }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < PER; x++) if (k[x] != 0xFFFF) order[base[k[x]] + r[x]] = i0 + 256u * x;
}

#ifndef DP_EMIT_Q
#define DP_EMIT_Q 6    // parked chains per chunk (16 B each in a global queue)
#endif
#ifndef DP_LINE
#define DP_LINE 8     // anchors per fetched line: 8 (32 B) measured best (2.04 ms; 16: 2.44 ms, 4: 2.06 ms) -- less LDS, one more wave per SIMD
#endif
// W = Wide (round 5): a run on 64-bit coordinates rings the LOW words.  The query coordinates of a chunk lie within one fragment of each other, so their low-word
// differences are the differences; on the reference side two anchors may be 2^32 (or 2^33, ...) apart, their low words a legal gap apart -- so the ring also keeps the
// high word of every anchor's strand-signed reference coordinate and a link counts only when the 64-bit difference has a zero high word (a subtract, a compare and an
// and per slot).  Everything else -- scores, components, the live-chain table, the emission -- is the 32-bit kernel's.  (Before: a wide run chained with the wave-per-
// chunk sweep kernel, which keeps band / 64 of its lanes busy: every genome forced wide, the headline's chaining took 2.5 x as long.)
template <int NB, int T, uint32_t DP_LDS_SLOTS, bool EXACT, class W = Narrow>   // EXACT: band == NB (the presets' bands), no per-slot band test
__global__ __launch_bounds__(T) void chain_dp_thread_kernel(uint32_t n_slots, const Chunk* chunks, const uint32_t* chunk_pair, const uint32_t* order, uint32_t band,
                                                            EmitCtxT<W> ec, unsigned long long* spill_best, uint32_t* spill_rr, uint4* emit_q, uint32_t emit_cap) {
    using Co = typename W::Co;
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
    uint32_t rh[W::wide ? NB : 1];                                                  // Wide: high word of the slot's strand-signed reference coordinate + 1
#pragma unroll
    for (int k = 0; k < NB; k++) { rq[k] = 0; rr[k] = 0; rs[k] = 0; rd[k] = 0; }
#pragma unroll
    for (int k = 0; k < (W::wide ? NB : 1); k++) rh[k] = 0;
    // Anchor fetch.  A lane walks its own chunk, so a plain per-lane load touches 64 different cache lines per instruction and
    // uses 4 bytes of each; with ~50k such streams per XCD the lines are evicted before their next element is wanted and every
    // anchor costs a 64-byte HBM fetch (measured: 15 GB read for 2.4 GB of anchors).  Instead every lane pulls whole lines
    // (DP_LINE anchors of one array) as 16-byte loads, one line ahead of use, and parks the current line in its own LDS
    // column [element][lane].  All lanes use the same element index: a lane's walk starts at its chunk's 64-byte-aligned
    // predecessor ("virtual" index v; elements before the chunk are skipped), which keeps the LDS reads conflict-free and
    // the refill branch wave-uniform.
    constexpr uint32_t LINE = DP_LINE;                       // anchors per fetched line (16 = 64 bytes)
    constexpr int APL = W::wide ? 2 : 4;                     // anchors per 16-byte load
    constexpr int LQ = LINE / APL;
    __shared__ uint32_t lds_q[LINE * T], lds_r[LINE * T];
    __shared__ uint32_t lds_rh[W::wide ? LINE * T : 1];     // Wide: the reference coordinates' high words (the query side needs its low words only)
    const uint32_t voff = ck.a_begin & (LINE - 1);
    const uint32_t vtot = n ? n + voff : 0;
    const Co* line_q = ec.anc_q + (ck.a_begin - voff); const Co* line_r = ec.anc_r + (ck.a_begin - voff);
    uint4 pq[LQ], pr[LQ];
#pragma unroll
    for (int j = 0; j < LQ; j++) { pq[j] = make_uint4(0, 0, 0, 0); pr[j] = make_uint4(0, 0, 0, 0); }
    if (vtot) {
#pragma unroll
        for (int j = 0; j < LQ; j++) { pq[j] = *(const uint4*)(line_q + APL * j); pr[j] = *(const uint4*)(line_r + APL * j); }
    }
    for (uint32_t v = 0;; v++) {
        const uint32_t kk = v & (LINE - 1);
        if (kk == 0) {                                                              // wave-uniform
            if (__ballot(v < vtot) == 0) break;
#pragma unroll
            for (int j = 0; j < LQ; j++) {
                if constexpr (W::wide) {                                            // two 64-bit anchors per load: (lo, hi, lo, hi)
                    lds_q[(2 * j + 0) * T + tid] = pq[j].x; lds_q[(2 * j + 1) * T + tid] = pq[j].z;
                    lds_r[(2 * j + 0) * T + tid] = pr[j].x; lds_r[(2 * j + 1) * T + tid] = pr[j].z;
                    lds_rh[(2 * j + 0) * T + tid] = pr[j].y; lds_rh[(2 * j + 1) * T + tid] = pr[j].w;
                } else {
                    lds_q[(4 * j + 0) * T + tid] = pq[j].x; lds_q[(4 * j + 1) * T + tid] = pq[j].y; lds_q[(4 * j + 2) * T + tid] = pq[j].z; lds_q[(4 * j + 3) * T + tid] = pq[j].w;
                    lds_r[(4 * j + 0) * T + tid] = pr[j].x; lds_r[(4 * j + 1) * T + tid] = pr[j].y; lds_r[(4 * j + 2) * T + tid] = pr[j].z; lds_r[(4 * j + 3) * T + tid] = pr[j].w;
                }
            }
            if (v + LINE < vtot) {
#pragma unroll
                for (int j = 0; j < LQ; j++) { pq[j] = *(const uint4*)(line_q + v + LINE + APL * j); pr[j] = *(const uint4*)(line_r + v + LINE + APL * j); }
            }
        }
        if (v < voff || v >= vtot) continue;
        const uint32_t i = v - voff;
        const uint2 a = make_uint2(lds_q[kk * T + tid], lds_r[kk * T + tid]);
        uint32_t q = a.x, r, sh = 0;                                                 // r, sh: low / high word of s = reverse ? ~(coordinate) : coordinate
        if constexpr (W::wide) {
            const uint32_t ah = lds_rh[kk * T + tid];
            const uint64_t co = (((uint64_t)ah << 32) | a.y) >> 1, sv = (a.y & 1u) ? ~co : co;
            r = (uint32_t)sv; sh = (uint32_t)(sv >> 32);
        } else r = (a.y & 1u) ? ~(a.y >> 1) : (a.y >> 1);
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
                        bool ok = (dq1 < BP_CHAIN_BAND) & (gap <= (uint32_t)MAX_GAP) & (sc > bscore);
                        if constexpr (W::wide) ok = ok & ((sh - rh[k]) == (r < rr[k] ? 1u : 0u));   // the 64-bit s_i - (s_j + 1) has a zero high word: dr1 IS the difference
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
        if constexpr (W::wide) {
#pragma unroll
            for (int k = NB - 1; k > 0; k--) rh[k] = rh[k - 1];
            rh[0] = sh + (r == 0xFFFFFFFFu ? 1u : 0u);                               // (s + 1)'s high word
        }
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
template <class W>
__global__ __launch_bounds__(256) void interval_emit_kernel(uint32_t n_slots, const Chunk* chunks, const unsigned long long* best, const uint32_t* chunk_pair, EmitCtxT<W> ec) {
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
