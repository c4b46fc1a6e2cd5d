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

#include "chain_types.h"
#include "chain_join.h"
#include "chain_chunk.h"
#include "chain_dp.h"
#include "chain_select.h"
#include "chain_stats.h"
#include "chain_estimate.h"

// ------------------------------------------------------------------------------------------------ host driver
namespace {

// per-genome inputs of the pair descriptors, gathered once per set (after its tables exist: the pointers do not move afterwards)
static void genome_halves(const skh_sketch_set* S) {
    std::lock_guard<std::mutex> lk(S->cache_mu);
    if (S->halves.size() == S->n_genomes && S->n_genomes) return;
    S->halves.resize(S->n_genomes);
    for (uint32_t g = 0; g < S->n_genomes; g++) {
        skh_sketch_set::GenomeHalf& h = S->halves[g];
        h.n_pos = (uint32_t)(S->pos_off[g + 1] - S->pos_off[g]); h.pos0 = (uint32_t)S->pos_off[g];
        h.seed = S->p_seed.p + S->pos_off[g]; h.g = S->p_g.p + S->pos_off[g]; h.rep = S->p_rep.p; h.salt = S->salt[g];
        h.ms = S->ms.p + S->ms_off[g]; h.tab = S->tab.p + S->tab_off[g]; h.nbk = S->n_buckets[g]; h.bmap = S->bmap.p + S->bmap_off[g];
        h.goff = S->d_goff.p + S->ctg_off[g] + g; h.host_goff = S->goff.data() + S->ctg_off[g] + g;
        h.g64 = nullptr; h.goff64 = nullptr; h.host_goff64 = nullptr;
        if (S->wide && S->wide_g[g]) {                                                // a genome beyond 31-bit coordinates: p_g holds its position indices
            h.g64 = S->p_g64.p + S->pos_off[g]; h.goff64 = S->d_goff64.p + S->ctg_off[g] + g; h.host_goff64 = S->goff64.data() + S->ctg_off[g] + g; h.goff = nullptr; h.host_goff = nullptr;
        }
        h.nctg = (uint32_t)(S->ctg_off[g + 1] - S->ctg_off[g]);
        h.total_len = S->total_len[g]; h.q10 = S->q10[g]; h.q50 = S->q50[g]; h.q90 = S->q90[g];
        const double cap = std::min(S->mean_ctg[g], 300000.);
        h.score_markers = ((double)(S->mk_off[g + 1] - S->mk_off[g]) * (double)S->params.c) * cap;
        h.score_len = (double)S->total_len[g] * cap;
        // chunks per contig <= len/20000 + 2 (every close advances the end point by 20000 inside the contig)
        h.chunk_bound = (uint32_t)(S->total_len[g] / CHUNK_SIZE + 2 * (uint64_t)h.nctg + 2);
    }
    S->lite.resize(S->n_genomes);
    for (uint32_t g = 0; g < S->n_genomes; g++) {
        const skh_sketch_set::GenomeHalf& h = S->halves[g];
        S->lite[g] = skh_sketch_set::HalfLite{h.score_markers, h.score_len, h.total_len, h.n_pos, h.nbk, h.nctg, h.chunk_bound, (uint8_t)(h.g64 != nullptr)};
    }
}

// the set's halves on the device (cached in the set; made from the host halves, which must exist)
static const GenomeDev* dev_halves(skh_ctx* ctx, const skh_sketch_set* S) {
    std::lock_guard<std::mutex> lk(S->cache_mu);
    if (!S->d_halves_ok) {
        std::vector<GenomeDev> v(S->n_genomes);
        for (uint32_t g = 0; g < S->n_genomes; g++) {
            const skh_sketch_set::GenomeHalf& h = S->halves[g]; GenomeDev& d = v[g];
            d.seed = h.seed; d.g = h.g; d.rep = h.rep; d.ms = h.ms; d.bmap = h.bmap; d.goff = h.goff; d.tab = h.tab;
            d.n_pos = h.n_pos; d.pos0 = h.pos0; d.nbk = h.nbk; d.salt = h.salt; d.nctg = h.nctg; d.pad = 0; d.total_len = h.total_len; d.q10 = h.q10; d.q50 = h.q50; d.q90 = h.q90; d.pad2 = 0;
        }
        S->d_halves.alloc(v.size() * sizeof(GenomeDev) + 16);
        h2d(S->d_halves.p, v.data(), v.size() * sizeof(GenomeDev), ctx->stream);
        if (v.size() * sizeof(GenomeDev) > ((size_t)1 << 20)) dsync(ctx->stream);     // (beyond the pinned ring the copy reads `v` itself)
        if (!S->d_halves_ev) S->d_halves_ev.reset(new DevEvent());
        S->d_halves_ev->record(ctx->stream); S->d_halves_stream = ctx->stream; S->d_halves_ok = true;
    } else if (S->d_halves_stream != ctx->stream) S->d_halves_ev->make_wait(ctx->stream);
    return (const GenomeDev*)S->d_halves.p;
}
static void drop_halves(const skh_sketch_set* S) { std::lock_guard<std::mutex> lk(S->cache_mu); S->halves.clear(); S->lite.clear(); S->d_halves_ok = false; }

// PairRec -> PairDesc: A = the enumerated side (the reference when switched), B = the probed side; the finalisation inputs by ref / query
__global__ __launch_bounds__(256) void expand_pairs_kernel(const PairRec* __restrict__ recs, uint32_t n, const GenomeDev* const* __restrict__ tabs /* the reference sets' tables, then the query sets' */,
                                                           uint32_t n_rsets, PairDesc* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const PairRec rec = recs[p];
    const GenomeDev hr = tabs[(rec.flags >> 8) & 0xFFFu][rec.r], hq = tabs[n_rsets + (rec.flags >> 20)][rec.q];
    const bool sw = (rec.flags & 4u) != 0;
    const GenomeDev& A = sw ? hr : hq; const GenomeDev& B = sw ? hq : hr;
    PairDesc pd;
    pd.a_seed = A.seed; pd.a_g = A.g; pd.a_rep = A.rep; pd.b_ms = B.ms; pd.b_tab = B.tab; pd.b_bmap = B.bmap; pd.a_goff = A.goff; pd.b_goff = B.goff;
    pd.a_n = (rec.flags & 8u) ? 0u : A.n_pos; pd.a_pos0 = A.pos0; pd.b_nbk = B.nbk; pd.b_salt = B.salt; pd.flags = rec.flags & 4u; pd.tile0 = rec.tile0;
    pd.a_nctg = A.nctg; pd.b_nctg = B.nctg; pd.nctg_q = hq.nctg; pd.nctg_r = hr.nctg; pd.ref_total_len = hr.total_len; pd.query_total_len = hq.total_len;
    pd.q10_q = hq.q10; pd.q50_q = hq.q50; pd.q90_q = hq.q90; pd.q10_r = hr.q10; pd.q50_r = hr.q50; pd.q90_r = hr.q90;
    out[p] = pd;
}

// same checksum as the oracle's ora_chain_stats.anchor_checksum: (query contig, query pos, ref contig, ref pos, reverse) per anchor
template <class Co, class Arr>
uint64_t fnv_anchors(const std::vector<Co>& anc, const std::vector<Co>& anc_r, size_t a0, size_t a1, const Arr& a_go, uint32_t a_n, const Arr& b_go, uint32_t b_n) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t x) { h ^= x; h *= 1099511628211ull; };
    for (size_t i = a0; i < a1; i++) {
        const Co gq = anc[i], gr = anc_r[i] >> 1; const uint32_t rev = (uint32_t)(anc_r[i] & 1u);
        const uint32_t qc = ctg_of(a_go, a_n, gq), rc = ctg_of(b_go, b_n, gr);
        mix(qc); mix((uint32_t)(gq - a_go[qc])); mix(rc); mix((uint32_t)(gr - b_go[rc])); mix(rev);
    }
    return h;
}

template <class T> T* upload(skh_ctx* ctx, const std::vector<T>& v) {
    T* d = ctx->arena.get<T>(v.size() ? v.size() : 1);
    h2d(d, v.data(), v.size() * sizeof(T), ctx->stream);
    return d;
}

}  // namespace

// the host's and the device's per-genome tables of a set whose seed tables exist, made ahead of the first chaining call (the sketch call does it while it waits for its last
// kernels: ~30 us of host work and an upload that would otherwise sit between the screen and the join)
// Round 5: made AHEAD, while the table build's kernels still run -- everything in the tables is known when the build is queued except a genome's salt (another one only
// when its seeds crowd a stretch of the hash range: rare) and, for a compact set, the place of its list storage; the build's finish says when either changed and the tables
// are made again.  (After the build's read-back the device is idle: 30-40 us per step.)
void prepare_halves(skh_ctx* ctx, const skh_sketch_set* S, bool ahead, bool again) {
    if (again) drop_halves(S);
    if ((S->tables_built || ahead) && S->n_genomes) { genome_halves(S); (void)dev_halves(ctx, S); }
}

// what the host keeps of a pair (the descriptor itself is made on the device: expand_pairs_kernel)
struct HostPair { uint32_t a_n, b_nbk, tile0, flags, a_nctg, b_nctg; };

// Slot table of the join kernels, written on the device from per-pair records: workgroup b runs the JOIN_GROUP tiles from (tile, pair) = slot[b] on; the tile groups of pairs
// [p0, p1) are dealt to eight queues by the probed sketch (queue = key % 8) and queue x owns slots x, x + 8, x + 16, ... (-> XCD x on MI355X), so
// all tiles probing one sketch run on the XCD whose L2 holds its table.  The host only computes each pair's first position in its queue.
__global__ __launch_bounds__(256) void slot_tile_kernel(uint32_t p0, uint32_t p1, const PairDesc* pairs, const uint32_t* queue_pos /* (p1-p0): queue << 29 | first */,
                                                        uint2* slot_tile) {
    const uint32_t p = p0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= p1) return;
    const uint32_t nt = (pairs[p].a_n + JOIN_TILE - 1) / JOIN_TILE, ngr = (nt + JOIN_GROUP - 1) / JOIN_GROUP, t0 = pairs[p].tile0, qp = queue_pos[p - p0];
    const uint32_t x = qp >> 29, first = qp & 0x1FFFFFFFu;
    for (uint32_t t = 0; t < ngr; t++) slot_tile[(size_t)(first + t) * 8 + x] = make_uint2(t0 + t * JOIN_GROUP, p);
}
// returns the device slot table for pairs [p0, p1) and its length
static uint2* xcd_slots(skh_ctx* ctx, uint32_t p0, uint32_t p1, const HostPair* pds, const PairDesc* d_pairs_all, const std::vector<uint32_t>& pair_key,
                        unsigned* n_slots) {
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> qp(p1 - p0);
    for (uint32_t p = p0; p < p1; p++) {
        const uint32_t x = pair_key[p] & 7u, nt = ((pds[p].a_n + JOIN_TILE - 1) / JOIN_TILE + JOIN_GROUP - 1) / JOIN_GROUP;   // tile groups = workgroups
        if (cnt[x] + nt >= (1u << 29)) throw Error("too many join tiles in one batch");
        qp[p - p0] = (x << 29) | cnt[x]; cnt[x] += nt;
    }
    uint32_t mx = 0; for (uint32_t v : cnt) mx = std::max(mx, v);
    *n_slots = mx * 8;
    uint2* d_slots = ctx->arena.get<uint2>((size_t)mx * 8 + 1);
    if (mx == 0) return d_slots;
    dfill(d_slots, 0xFF, (size_t)mx * 8 * sizeof(uint2), ctx->stream);
    uint32_t* d_qp = ctx->arena.get<uint32_t>(qp.size()); h2d(d_qp, qp.data(), qp.size() * 4, ctx->stream);
    SKH_LAUNCH(slot_tile_kernel, (p1 - p0 + 255) / 256, 256, 0, ctx->stream, p0, p1, d_pairs_all, (const uint32_t*)d_qp, d_slots);
    check_launch("slot_tile");
    return d_slots;
}

// Wide runs: the join's anchors (32-bit records: a coordinate, or -- on the side of a wide set -- a position index) -> 64-bit coordinates.
// One thread per anchor; its pair by a search in the batch's anchor prefix.
__global__ __launch_bounds__(256) void widen_anchors_kernel(uint32_t n_anchors, uint32_t n_pairs, const uint32_t* pa0, const WidePair* wide, const uint32_t* anc_q32, const uint32_t* anc_r32,
                                                            uint64_t* anc_q, uint64_t* anc_r) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_anchors) return;
    uint32_t lo = 0, hi = n_pairs;                                                  // largest p with pa0[p] <= i
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pa0[mid] <= i) lo = mid; else hi = mid; }
    const WidePair wp = wide[lo];
    const uint32_t q = anc_q32[i], r = anc_r32[i];
    anc_q[i] = wp.a_is_index ? wp.a_g[q] >> 1 : (uint64_t)q;
    anc_r[i] = wp.b_g64 ? (wp.b_g64[r >> 1] & ~1ull) | (r & 1u) : (uint64_t)r;
}

namespace {

struct ChainJob {                                        // what one run over a list of pairs needs (chain_pairs fills it)
    PairRec* recs = nullptr; uint32_t n_pairs = 0;       // in the context's pinned buffer: copied to the device as they are, expanded there
    std::vector<HostPair> hp;                            // the host's view of the same pairs
    std::vector<const GenomeDev*> tabs; uint32_t n_rsets = 0;   // device tables of the reference sets, then of the query sets
    std::vector<WidePair> wps;                           // a run with a wide sketch set: one per pair
    std::vector<CoArr> host_go_a64, host_go_b64;         //   (host tables for the stats' checksum)
    std::vector<uint32_t> chunk_bound, pair_key;
    std::vector<const uint32_t*> host_go_a, host_go_b;   // only with stats
    uint32_t c = 0, k = 0, band = 0;
    const GbdtModel* model = nullptr;
    skh_map_params mp{};
};

// Runs the chaining pipeline over all pairs of the job.  W: the run's coordinate width (chain_types.h).
template <class W>
void chain_run(skh_ctx* ctx, ChainJob& job, skh_ani_result* out, skh_chain_stats* stats) {
    using Co = typename W::Co;
    // (A join that walks all tiles of a pair in one workgroup and writes the anchors in one pass -- no probe records, no pair counts on the host -- was
    //  built and measured in round 2: 4.7 ms against 3.4 ms for count + fill; long-lived workgroups hide the probe latency worse.  DESIGN.md section 5.)
    HostPair* pds = job.hp.data();
    const uint32_t NP = job.n_pairs, band = job.band, c = job.c, k = job.k;
    const GbdtModel* model = job.model; const skh_map_params& mp = job.mp;
    StageTrace tr(ctx);
    const bool join_trace = getenv("SKH_TRACE_JOIN") != nullptr;
    uint64_t n_tiles_all = 0;
    for (uint32_t p = 0; p < NP; p++) {
        pds[p].tile0 = job.recs[p].tile0 = (uint32_t)n_tiles_all;
        n_tiles_all += (pds[p].a_n + JOIN_TILE - 1) / JOIN_TILE;
        if (n_tiles_all >= 0xFFFFFFF0ull) throw std::invalid_argument("too many sketch positions in one chain call; split the pair list");
    }
    const uint32_t NT = (uint32_t)n_tiles_all;
    PairDesc* d_pairs_all = ctx->arena.get<PairDesc>(NP ? NP : 1);
    {   // 16 bytes per pair go up; the descriptors are made from the sets' resident per-genome tables
        PairRec* d_recs = ctx->arena.get<PairRec>(NP ? NP : 1);
        h2d_big(d_recs, job.recs, (size_t)NP * sizeof(PairRec), ctx->stream);
        const GenomeDev* const* d_tabs = upload(ctx, job.tabs);
        SKH_LAUNCH(expand_pairs_kernel, (NP + 255) / 256, 256, 0, ctx->stream, (const PairRec*)d_recs, NP, d_tabs, job.n_rsets, d_pairs_all);
        check_launch("expand_pairs");
    }
    const WidePair* d_wide_all = nullptr;
    if (W::wide) d_wide_all = upload(ctx, job.wps);
    skh_ani_result* d_out = ctx->arena.get<skh_ani_result>(NP);
    uint32_t* d_err = ctx->arena.get<uint32_t>(1);
    uint32_t* tile_anch = ctx->arena.get<uint32_t>((size_t)NT + 1); uint32_t* tile_hits = ctx->arena.get<uint32_t>((size_t)NT + 1);
    uint32_t* d_pair_anch = ctx->arena.get<uint32_t>(NP); uint32_t* d_pair_inq = ctx->arena.get<uint32_t>(NP);
    { FillRegions fr; fr.add(d_err, 1, 0u); fr.add(d_pair_anch, NP, 0u); fr.add(d_pair_inq, NP, 0u); fill_regions(ctx, fr); }

    const uint64_t ANCH_BUDGET = ctx->tune.chain_anchors;        // anchors per batch (~30 B of scratch each)
    const uint32_t SUPER_TILES = ctx->tune.chain_super_tiles;    // join tiles per count pass (up to 8 KiB of hit records each)
    const uint32_t big_min = std::max<uint32_t>(2, std::min<uint32_t>(ctx->tune.greedy_big_min, GREEDY_LDS + 1));   // candidate intervals from which a pair takes greedy_big_kernel
    std::vector<uint32_t> pair_anch(NP), pair_inq(NP);
    uint32_t sp0 = 0;
    while (sp0 < NP) {
        // ---- super-batch: pairs [sp0, sp1) = tiles [st0, st1); the count pass records one probe result per position
        uint32_t sp1 = sp0;
        while (sp1 < NP && (sp1 == sp0 || (sp1 + 1 < NP ? pds[sp1 + 1].tile0 : NT) - pds[sp0].tile0 <= SUPER_TILES)) sp1++;
        const uint32_t st0 = pds[sp0].tile0, st1 = sp1 < NP ? pds[sp1].tile0 : NT, snt = st1 - st0;
        const std::vector<size_t> super_mark = ctx->arena.mark();
        uint2* hit_rec = ctx->arena.get<uint2>((size_t)snt * JOIN_TILE + 1);          // a tile's hit records: up to one per position, written densely from the tile's start
        unsigned long long* inq_mask = ctx->arena.get<unsigned long long>((size_t)snt * (JOIN_TILE / 64) + 1);
        // kernels index tiles globally: shift the record arrays so that tile st0 maps to their start
        uint2* pis = hit_rec - (size_t)st0 * JOIN_TILE; unsigned long long* imk = inq_mask - (size_t)st0 * (JOIN_TILE / 64);
        uint32_t bm_words = 0;                                                     // LDS for the largest bitmap of the batch, up to 32 KB
        for (uint32_t p = sp0; p < sp1; p++) bm_words = std::max(bm_words, ((pds[p].b_nbk + TAB_FILTER_HOMES - 1) / TAB_FILTER_HOMES + 3) / 4 * 4);
        if (bm_words > ctx->tune.join_bitmap_words) bm_words = ctx->tune.join_bitmap_words;   // pairs with a larger bitmap probe the table directly
        uint2* d_super_slots = nullptr; unsigned n_super_slots = 0;            // reused by the fill pass when the batch is the whole super-batch
        if (snt) {
            d_super_slots = xcd_slots(ctx, sp0, sp1, pds, d_pairs_all, job.pair_key, &n_super_slots);
            const size_t lds = (size_t)bm_words * 4 + (size_t)JOIN_GROUP * JOIN_Q * 8;   // the filter + a probe queue per wave
            if (lds > ((size_t)48 << 10)) kernel_allow_lds(join_count_kernel<false>, lds);
            if (!join_trace)
                SKH_LAUNCH(join_count_kernel<false>, n_super_slots, JOIN_THREADS, lds, ctx->stream, (const PairDesc*)d_pairs_all, (const uint2*)d_super_slots,
                           band, tile_anch, tile_hits, d_pair_anch, d_pair_inq, pis, imk, bm_words, (unsigned long long*)nullptr);
            else {                                                                     // SKH_TRACE_JOIN=1: where a wave of the count pass spends its cycles (slower: every phase waits for its loads)
                unsigned long long* d_prof = ctx->arena.get<unsigned long long>(16); dzero(d_prof, 128, ctx->stream);
                if (lds > ((size_t)48 << 10)) kernel_allow_lds(join_count_kernel<true>, lds);
                SKH_LAUNCH(join_count_kernel<true>, n_super_slots, JOIN_THREADS, lds, ctx->stream, (const PairDesc*)d_pairs_all, (const uint2*)d_super_slots,
                           band, tile_anch, tile_hits, d_pair_anch, d_pair_inq, pis, imk, bm_words, d_prof);
                unsigned long long hp[16]; d2h(hp, d_prof, 128, ctx->stream);
                const double nw = hp[8] ? (double)hp[8] : 1.;
                fprintf(stderr, "[skh trace] join_count: %llu waves; cycles per wave: wait for the round's hashes %.0f, filter + prefix %.0f, queue %.0f, wait for the home slots %.0f, "
                        "cluster walks %.0f (%.1f steps), results back %.0f, classify + list heads + stores %.0f\n", hp[8], hp[0] / nw, hp[1] / nw, hp[2] / nw, hp[3] / nw, hp[4] / nw,
                        hp[7] / nw, hp[5] / nw, hp[6] / nw);
            }
            check_launch("join_count");
        }
        tr.mark("join_count (+slots)");
        // the tile scan of the whole super-batch is queued before the pair counts are waited for: it is what the fill pass needs when the anchors of all
        // these pairs fit one batch (the usual case), and it runs while the host waits
        uint32_t* toff_super = ctx->arena.get<uint32_t>(snt + 1);
        exclusive_scan_u32(ctx, tile_anch + st0, snt, toff_super);
        d2h(pair_anch.data() + sp0, d_pair_anch + sp0, (size_t)(sp1 - sp0) * 4, ctx->stream);
        if (stats) d2h(pair_inq.data() + sp0, d_pair_inq + sp0, (size_t)(sp1 - sp0) * 4, ctx->stream);
        tr.mark("d2h pair counts");
        uint32_t p0 = sp0;
        while (p0 < sp1) {
        uint64_t na = 0; uint32_t p1 = p0;
        while (p1 < sp1 && (p1 == p0 || na + pair_anch[p1] <= ANCH_BUDGET)) { na += pair_anch[p1]; p1++; }
        const uint32_t np = p1 - p0;
        const std::vector<size_t> arena_mark = ctx->arena.mark();
        if (na >= 0xFFFFFFF0ull) throw Error("a single genome pair produces more than 2^32 anchors");
        // the fill pass needs the batch's anchor count and the tile scan only: it is queued first, and the per-pair prefix arrays of the later stages
        // (a host loop over the pairs + four uploads, ~0.1 ms) are made while it runs
        const uint32_t NA = (uint32_t)na;
        const uint32_t t0 = pds[p0].tile0, t1 = p1 < NP ? pds[p1].tile0 : NT, nt = t1 - t0;
        const PairDesc* d_pairs = d_pairs_all + p0; const WidePair* d_wide = W::wide ? d_wide_all + p0 : nullptr;
        uint32_t* anc_q32 = ctx->arena.get<uint32_t>((size_t)NA + 16); uint32_t* anc_r32 = ctx->arena.get<uint32_t>((size_t)NA + 16);   // what the join writes
        uint32_t* toff_a = toff_super;
        if (t0 != st0 || t1 != st1) { toff_a = ctx->arena.get<uint32_t>(nt + 1); exclusive_scan_u32(ctx, tile_anch + t0, nt, toff_a); }
        tr.mark("tile scan");
        if (nt) {
            uint2* d_slots = d_super_slots; unsigned n_slots = n_super_slots;
            if (t0 != st0 || t1 != st1) d_slots = xcd_slots(ctx, p0, p1, pds, d_pairs_all, job.pair_key, &n_slots);
            SKH_LAUNCH(join_fill_kernel, n_slots, JOIN_THREADS, 0, ctx->stream, (const PairDesc*)d_pairs_all, (const uint2*)d_slots,
                       t0, (const uint32_t*)toff_a, (const uint32_t*)tile_hits, (const uint2*)pis, anc_q32, anc_r32);
            check_launch("join_fill");
        }
        // per-pair prefix arrays (batch-relative): anchors, chunks, candidate intervals, global sort scratch of the fallback selection kernel
        std::vector<uint32_t> pa0(np + 1, 0), pc0(np + 1, 0), pi0(np + 1, 0), ps0(np + 1, 0);
        for (uint32_t i = 0; i < np; i++) {
            const uint32_t an = pair_anch[p0 + i];
            pa0[i + 1] = pa0[i] + an;
            pc0[i + 1] = pc0[i] + (an ? std::min(job.chunk_bound[p0 + i], an) : 0);
            const uint32_t icap = an / MIN_ANCHORS;
            pi0[i + 1] = pi0[i] + icap; ps0[i + 1] = ps0[i] + (icap >= big_min ? icap : 0);   // candidate slots of greedy_big_kernel's scratch
            if (ps0[i + 1] < ps0[i]) throw Error("too many candidate intervals in one chain batch");
        }
        const uint32_t NC = pc0[np], NI = pi0[np], NS = ps0[np];
        // one upload for the four arrays (each copy behind the fill pass is a launch of its own)
        std::vector<uint32_t> pfx(4 * ((size_t)np + 1));
        memcpy(pfx.data(), pa0.data(), ((size_t)np + 1) * 4); memcpy(pfx.data() + (np + 1), pc0.data(), ((size_t)np + 1) * 4);
        memcpy(pfx.data() + 2 * ((size_t)np + 1), pi0.data(), ((size_t)np + 1) * 4); memcpy(pfx.data() + 3 * ((size_t)np + 1), ps0.data(), ((size_t)np + 1) * 4);
        uint32_t* d_pa0 = upload(ctx, pfx); uint32_t* d_pc0 = d_pa0 + (np + 1); uint32_t* d_pi0 = d_pc0 + (np + 1); uint32_t* d_ps0 = d_pi0 + (np + 1);
        tr.mark("join_fill (+slots)");
        Co *anc_q, *anc_r;
        if constexpr (W::wide) {
            anc_q = ctx->arena.get<uint64_t>((size_t)NA + 16); anc_r = ctx->arena.get<uint64_t>((size_t)NA + 16);
            if (NA) {
                SKH_LAUNCH(widen_anchors_kernel, (NA + 255) / 256, 256, 0, ctx->stream, NA, np, (const uint32_t*)d_pa0, d_wide, (const uint32_t*)anc_q32, (const uint32_t*)anc_r32, anc_q, anc_r);
                check_launch("widen_anchors");
            }
        } else { anc_q = anc_q32; anc_r = anc_r32; }
        Chunk* chunks = ctx->arena.get<Chunk>(NC + 1); uint32_t* chunk_pair = ctx->arena.get<uint32_t>(NC + 1);
        uint32_t* n_chunks = ctx->arena.get<uint32_t>(np);
        SKH_LAUNCH(chunk_kernel<W>, (np + 3) / 4, 256, 0, ctx->stream, np, d_pairs, d_wide, (const uint32_t*)d_pa0, (const uint32_t*)(d_pair_anch + p0),
                   (const uint32_t*)d_pc0, (const Co*)anc_q, chunks, chunk_pair, n_chunks, d_err);
        check_launch("chunk");
        tr.mark("chunk");
        Interval* ivls = ctx->arena.get<Interval>(NI + 1); uint32_t* ivl_cnt = ctx->arena.get<uint32_t>(np);
        uint32_t* ivl_next = ctx->arena.get<uint32_t>(NI + 1); uint32_t* greedy_scratch = ctx->arena.get<uint32_t>((size_t)NS * GREEDY_BIG_WORDS + 1);
        uint32_t* chunk_head = ctx->arena.get<uint32_t>(NC + 1); uint32_t* n_acc = ctx->arena.get<uint32_t>(np);
        FillRegions stage_fills; stage_fills.add(ivl_cnt, np, 0u); stage_fills.add(chunk_head, (uint64_t)NC + 1, 0xFFFFFFFFu);   // (+ the DP order's histogram below: one launch)
        bool stage_filled = false;
        const EmitCtxT<W> ec{anc_q, anc_r, d_pairs, d_wide, d_pc0, d_pi0, ivl_cnt, ivls, d_err};
        if (NC) {
            bool chained = false;
            if (band <= (W::wide ? 40u : 84u) && !(W::wide && ctx->tune.wide_sweep_dp)) {   // fused thread-per-chunk chaining + interval emission (64-bit coordinates: rings up to 40 slots, chain_dp.h)
                chained = true;
                constexpr int T = 64;
                const unsigned gt = (NC + T - 1) / T;
                const uint32_t ls = ctx->tune.chain_dp_lds_slots <= 1 ? 1u : 8u;  // 1: tests push every second live chain through the spill table (4 / 6 slots measured: 1.65 / 1.55 vs 1.53 ms)
                const size_t n_spill = band + 1 > ls ? (size_t)(band + 1 - ls) * gt * T : 1;   // written only by chunks with more live chains than LDS slots
                unsigned long long* spill_best = ctx->arena.get<unsigned long long>(n_spill); uint32_t* spill_rr = ctx->arena.get<uint32_t>(n_spill);
                uint4* emit_q = ctx->arena.get<uint4>((size_t)DP_EMIT_Q * gt * T);
                const unsigned ob = (NC + DP_ORDER_BLOCK - 1) / DP_ORDER_BLOCK;
                uint32_t* order = ctx->arena.get<uint32_t>(NC); uint32_t* ohist = ctx->arena.get<uint32_t>(2 * DP_ORDER_KEYS);   // histogram | scatter cursors
                stage_fills.add(ohist, 2 * DP_ORDER_KEYS, 0u); fill_regions(ctx, stage_fills); stage_filled = true;
                SKH_LAUNCH(dp_order_hist_kernel, ob, 256, 0, ctx->stream, NC, (const Chunk*)chunks, ohist);
                SKH_LAUNCH(dp_order_scatter_kernel, ob, 256, 0, ctx->stream, NC, (const Chunk*)chunks, (const uint32_t*)ohist, ohist + DP_ORDER_KEYS, order);
                check_launch("dp_order");
#define SKH_DPT2(NB, LS, EX) SKH_LAUNCH((chain_dp_thread_kernel<NB, T, LS, EX, W>), gt, T, 0, ctx->stream, NC, (const Chunk*)chunks, (const uint32_t*)chunk_pair, (const uint32_t*)order, band, ec, spill_best, spill_rr, emit_q, ls == 1 ? 1u : (uint32_t)DP_EMIT_Q)   /* test mode: queue of one, the rest written directly */
#define SKH_DPT(NB, EX) do { if (ls == 1) SKH_DPT2(NB, 1, EX); else SKH_DPT2(NB, 8, EX); } while (0)
                // the presets' bands (2500 / c for c = 200, 125, 70, 30) get kernels with exactly that many ring slots
                if (band == 12) SKH_DPT(12, true); else if (band == 20) SKH_DPT(20, true); else if (band == 35) SKH_DPT(35, true);
                else if (band <= 12) SKH_DPT(12, false); else if (band <= 20) SKH_DPT(20, false); else if (band <= 28) SKH_DPT(28, false); else if (band <= 40) SKH_DPT(40, false);
                else if constexpr (!W::wide) { if (band == 83) SKH_DPT(83, true); else SKH_DPT(84, false); }
#undef SKH_DPT
#undef SKH_DPT2
                check_launch("chain_dp_thread");
            }
            if (!chained) {     // wave-per-chunk sweep + per-anchor argmax records + emit
                fill_regions(ctx, stage_fills); stage_filled = true;
                unsigned long long* best = ctx->arena.get<unsigned long long>((size_t)NA + 64);
                dzero(best, ((uint64_t)NA + 64) * 8, ctx->stream);
                const unsigned gb = (NC + 3) / 4;
                uint32_t* dp_state = band > 256 ? ctx->arena.get<uint32_t>(3 * (size_t)NA + 4) : nullptr;   // c < 10: earlier anchors' state goes through memory
#define SKH_DP(PB) SKH_LAUNCH((chain_dp_kernel<PB, W>), gb, 256, 0, ctx->stream, NC, (const Chunk*)chunks, band, (const Co*)anc_q, (const Co*)anc_r, best, dp_state)
                if (band <= 64) SKH_DP(1); else if (band <= 128) SKH_DP(2); else if (band <= 192) SKH_DP(3); else if (band <= 256) SKH_DP(4); else SKH_DP(0);
#undef SKH_DP
                check_launch("chain_dp");
                SKH_LAUNCH(interval_emit_kernel<W>, (NC + 3) / 4, 256, 0, ctx->stream, NC, (const Chunk*)chunks, (const unsigned long long*)best,
                           (const uint32_t*)chunk_pair, ec);
                check_launch("interval_emit");
            }
        }
        if (!stage_filled) fill_regions(ctx, stage_fills);
#define SKH_GREEDY(CAP) SKH_LAUNCH(greedy_fast_kernel<CAP>, np, 64, 0, ctx->stream, np, (const uint32_t*)g_order, (const uint32_t*)d_pi0, (const uint32_t*)d_pc0, \
                   (const uint32_t*)ivl_cnt, (const Interval*)ivls, ctx->tune.greedy_len_limit, big_min, ivl_next, chunk_head, n_acc); check_launch("greedy_fast")
        tr.mark("dp (+order sort)");
        uint32_t* g_keys = ctx->arena.get<uint32_t>(np); uint32_t* g_order = ctx->arena.get<uint32_t>(np);
        if (np <= GREEDY_ORDER_ONE_MAX) {                                             // one launch (it was six: keys + a radix sort of 9,500 values)
            SKH_LAUNCH(greedy_order_kernel, 1u, 1024, 0, ctx->stream, np, (const uint32_t*)ivl_cnt, g_order);
            check_launch("greedy_order");
        } else {
            SKH_LAUNCH(greedy_order_keys_kernel, (np + 255) / 256, 256, 0, ctx->stream, np, (const uint32_t*)ivl_cnt, g_keys, g_order);
            check_launch("greedy_order_keys");
            sort_pairs_u32_u32(ctx, g_keys, g_order, np, 16);
        }
        SKH_GREEDY(256); SKH_GREEDY(512); SKH_GREEDY(1024);
#undef SKH_GREEDY
        SKH_LAUNCH(greedy_kernel, (np + 3) / 4, 256, 0, ctx->stream, np, (const uint32_t*)d_pi0, (const uint32_t*)d_pc0,
                   (const uint32_t*)ivl_cnt, (const Interval*)ivls, big_min, ivl_next, chunk_head, n_acc);
        check_launch("greedy");
        if (NS) {                                                                     // the batch has a pair that may reach big_min candidates
            SKH_LAUNCH(greedy_big_kernel, np, 1024, 0, ctx->stream, np, (const uint32_t*)d_pi0, (const uint32_t*)d_ps0, (const uint32_t*)d_pc0,
                       (const uint32_t*)ivl_cnt, (const Interval*)ivls, big_min, greedy_scratch, ivl_next, chunk_head, n_acc);
            check_launch("greedy_big");
        }
        tr.mark("greedy");
        double* chunk_est = ctx->arena.get<double>(NC + 1); uint32_t* chunk_w = ctx->arena.get<uint32_t>(NC + 1);
        uint4* chunk_sums = ctx->arena.get<uint4>(NC + 1);
        if (NC) {
            SKH_LAUNCH(chunk_stats_kernel<W>, (NC + 255) / 256, 256, 0, ctx->stream, NC, (const Chunk*)chunks, (const uint32_t*)chunk_pair, (const uint32_t*)chunk_head,
                       (const uint32_t*)ivl_next, (const Interval*)ivls, d_pairs, d_wide, (const unsigned long long*)imk, c, k, chunk_est, chunk_w, chunk_sums);
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
#define SKH_FIN(CAP, MIN, MAX, THR) SKH_LAUNCH((finalize_kernel<CAP, MIN, MAX, THR>), np, THR, 0, ctx->stream, fa, d_pairs, (const uint32_t*)d_pc0, (const uint32_t*)n_chunks, \
                   (const double*)chunk_est, (const uint32_t*)chunk_w, (const uint4*)chunk_sums, fs, n_est, d_out + p0); \
        check_launch("finalize")
        bool any_mid = false, any_long = false;                                       // a pair that may have more than 320 / 1024 chunks (the latter: the global-memory instantiation)
        for (uint32_t i = 0; i < np && !any_long; i++) { any_mid = any_mid || pc0[i + 1] - pc0[i] > 320; any_long = pc0[i + 1] - pc0[i] > 1024; }
        SKH_FIN(320, 0, 320, FIN_THREADS);
        if (any_mid) { SKH_FIN(1024, 321, 1024, FIN_THREADS); }
        if (any_long) { SKH_FIN(1, 1025, 0xFFFFFFFFu, 1024); }
#undef SKH_FIN
        tr.mark("finalize");
        if (stats) {   // parity/debug path: pull the stage sizes (and the anchors, for the checksum) back to the host
            std::vector<uint32_t> h_nc(np), h_ni(np), h_nacc(np), h_ne(np);
            d2h(h_nc.data(), n_chunks, np * 4, ctx->stream); d2h(h_ni.data(), ivl_cnt, np * 4, ctx->stream);
            d2h(h_nacc.data(), n_acc, np * 4, ctx->stream); d2h(h_ne.data(), n_est, np * 4, ctx->stream);
            std::vector<Co> hanc((size_t)NA), hanr((size_t)NA);
            d2h(hanc.data(), anc_q, (uint64_t)NA * sizeof(Co), ctx->stream); d2h(hanr.data(), anc_r, (uint64_t)NA * sizeof(Co), ctx->stream);
            for (uint32_t i = 0; i < np; i++) {
                skh_chain_stats& st = stats[p0 + i];
                const uint32_t an = std::min(pair_anch[p0 + i], pa0[i + 1] - pa0[i]);
                st.switched = (pds[p0 + i].flags >> 2) & 1u; st.n_chunks = h_nc[i]; st.n_intervals = h_ni[i]; st.n_accepted = h_nacc[i]; st.n_estimates = h_ne[i];
                st.reserved = 0; st.n_anchors = pair_anch[p0 + i]; st.n_qpos = pair_inq[p0 + i];
                if constexpr (W::wide) st.anchor_checksum = an ? fnv_anchors(hanc, hanr, pa0[i], pa0[i] + an, job.host_go_a64[p0 + i], pds[p0 + i].a_nctg, job.host_go_b64[p0 + i], pds[p0 + i].b_nctg) : 0;
                else st.anchor_checksum = an ? fnv_anchors(hanc, hanr, pa0[i], pa0[i] + an, job.host_go_a[p0 + i], pds[p0 + i].a_nctg, job.host_go_b[p0 + i], pds[p0 + i].b_nctg) : 0;
                if (pair_anch[p0 + i] == 0) { st.switched = 1; st.n_qpos = 0; }   // reference returns (default, true) when there are no anchors (chain.rs:619,719)
            }
        }
        // the batch's results; this is also the batch's synchronisation (skh_triangle hands a pinned array: one DMA; a caller's own array: through the pinned ring, dev.h d2h)
        if (ctx->pin_results.holds(out + p0)) d2h_pinned(out + p0, d_out + p0, (size_t)np * sizeof(skh_ani_result), ctx->stream);
        else d2h(out + p0, d_out + p0, (size_t)np * sizeof(skh_ani_result), ctx->stream);
        ctx->arena.rewind(arena_mark);
        p0 = p1;
        }
        ctx->arena.rewind(super_mark);
        sp0 = sp1;
    }
    uint32_t h_err = 0;
    d2h(&h_err, d_err, 4, ctx->stream);
    tr.mark("results d2h");
    if (h_err) throw Error("internal capacity bound violated in chain pipeline (" + std::to_string(h_err) + " events)");
}

}  // namespace

void chain_pairs(skh_ctx* ctx, const skh_sketch_set* const* Rsets, uint32_t n_rsets, const uint32_t* pair_rset, const skh_sketch_set* const* Qsets, uint32_t n_qsets,
                 const uint32_t* pair_qset, const uint32_t* pair_ref, const uint32_t* pair_query, uint64_t n_pairs_all, const skh_map_params& mp, skh_ani_result* out,
                 skh_chain_stats* stats, bool tie_by_rank, const std::function<void()>* tables_pending) {
    if (n_pairs_all == 0) { if (tables_pending) (*tables_pending)(); return; }
    if (n_pairs_all > 0x7FFFFFFFull) throw std::invalid_argument("too many pairs in one call");
    if (n_rsets == 0 || !Rsets[0]) throw std::invalid_argument("no reference sketch set");
    if (n_qsets == 0 || !Qsets[0]) throw std::invalid_argument("no query sketch set");
    ChainJob job;
    job.c = Qsets[0]->params.c; job.k = Qsets[0]->params.k; job.mp = mp;
    for (uint32_t x = 0; x < n_rsets; x++)
        if (!Rsets[x] || Rsets[x]->params.c != job.c || Rsets[x]->params.k != job.k) throw std::invalid_argument("ref and query sketches were built with different c/k");
    for (uint32_t x = 0; x < n_qsets; x++)
        if (!Qsets[x] || Qsets[x]->params.c != job.c || Qsets[x]->params.k != job.k) throw std::invalid_argument("ref and query sketches were built with different c/k");
    if (!tables_pending) {
        for (uint32_t x = 0; x < n_rsets; x++) ensure_tables(ctx, Rsets[x]);        // (sets created with deferred tables)
        for (uint32_t x = 0; x < n_qsets; x++) ensure_tables(ctx, Qsets[x]);
    }
    bool pending_done = false;
    struct PendingGuard { const std::function<void()>* f; bool* done; ~PendingGuard() { if (f && !*done) { try { (*f)(); } catch (...) {} } } } pending_guard{tables_pending, &pending_done};   // never left queued
    job.band = BP_CHAIN_BAND / job.c;                                               // chain.rs:111-112 index_chain_band (ref sketch's c)
    if (mp.learned_ani) {
        job.model = std::abs((int)job.c - 125) < std::abs((int)job.c - 200) ? &ctx->model_c125 : &ctx->model_c200;   // regression.rs:15-22
        if (!job.model->loaded()) throw std::invalid_argument("learned_ani requested but skh_load_models was not called");
    }
    const uint32_t NP = (uint32_t)n_pairs_all;
    StageTrace tr(ctx);
    // ---- pair descriptors, from the sets' per-genome halves
    for (uint32_t x = 0; x < n_rsets; x++) genome_halves(Rsets[x]);
    for (uint32_t x = 0; x < n_qsets; x++) genome_halves(Qsets[x]);
    auto halves_of = [&](uint32_t p, const skh_sketch_set*& R, const skh_sketch_set*& Q, uint32_t& rs, uint32_t& qs) {
        rs = pair_rset ? pair_rset[p] : 0u; qs = pair_qset ? pair_qset[p] : 0u;
        if (rs >= n_rsets || qs >= n_qsets) throw std::invalid_argument("pair names a sketch set that was not passed");
        R = Rsets[rs]; Q = Qsets[qs];
        if (pair_ref[p] >= R->n_genomes || pair_query[p] >= Q->n_genomes) throw std::invalid_argument("pair index out of range");
    };
    // A pair with a genome beyond 31-bit coordinates (internal.h: wide sets) is chained on 64-bit coordinates from the chunking on; the other pairs of
    // the call -- all of them, unless a set is wide -- take the ordinary run.  Two runs then, each over its own list of the call's pairs.
    bool any_wide_set = false;
    for (uint32_t x = 0; x < n_rsets; x++) any_wide_set = any_wide_set || Rsets[x]->wide;
    for (uint32_t x = 0; x < n_qsets; x++) any_wide_set = any_wide_set || Qsets[x]->wide;
    std::vector<uint32_t> sel[2];                                                     // [0] ordinary pairs, [1] pairs with a wide genome (only filled when there are any)
    if (any_wide_set) {
        for (uint32_t p = 0; p < NP; p++) {
            const skh_sketch_set *R, *Q; uint32_t rs, qs; halves_of(p, R, Q, rs, qs);
            sel[(R->lite[pair_ref[p]].wide || Q->lite[pair_query[p]].wide) ? 1 : 0].push_back(p);
        }
        if (sel[1].empty()) sel[0].clear();                                            // nothing wide after all: one run over the call's pairs as they are
    }
    const bool split = !sel[1].empty();
    if (n_rsets > PAIR_MAX_SETS || n_qsets > PAIR_MAX_SETS) throw std::invalid_argument("more than 4096 sketch sets on one side of a chaining call");
    auto device_tables = [&] {                                                        // the sets' per-genome tables on the device: reference sets, then query sets
        job.tabs.clear(); job.n_rsets = n_rsets;
        for (uint32_t x = 0; x < n_rsets; x++) job.tabs.push_back(dev_halves(ctx, Rsets[x]));
        for (uint32_t x = 0; x < n_qsets; x++) job.tabs.push_back(dev_halves(ctx, Qsets[x]));
    };
    device_tables();
    for (int run = 0; run < (split ? 2 : 1); run++) {
        const bool wide_run = split && run == 1;
        const uint32_t* idx = split ? sel[run].data() : nullptr;
        const uint32_t n = split ? (uint32_t)sel[run].size() : NP;
        if (n == 0) continue;
        job.recs = (PairRec*)ctx->pin_pairs.need((size_t)n * sizeof(PairRec)); job.n_pairs = n; job.hp.assign(n, HostPair{});
        job.chunk_bound.assign(n, 0); job.pair_key.assign(n, 0);
        job.wps.clear(); job.host_go_a.clear(); job.host_go_b.clear(); job.host_go_a64.clear(); job.host_go_b64.clear();
        if (wide_run) job.wps.resize(n);
        if (stats) { job.host_go_a.resize(n); job.host_go_b.resize(n); if (wide_run) { job.host_go_a64.resize(n); job.host_go_b64.resize(n); } }
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t p = idx ? idx[i] : i;
            const skh_sketch_set *R, *Q; uint32_t rs, qs; halves_of(p, R, Q, rs, qs);
            const uint32_t r = pair_ref[p], q = pair_query[p];
            const skh_sketch_set::HalfLite& hr = R->lite[r]; const skh_sketch_set::HalfLite& hq = Q->lite[q];
            const bool empty = hr.nctg == 0 || hq.nctg == 0;                          // chain.rs:618-620
            // chain.rs:15-26 switch_qr with the inputs of chain.rs:625-649
            const bool both_long = hq.total_len > 100000 && hr.total_len > 100000;
            const double sq = both_long ? hq.score_markers : hq.score_len, sr = both_long ? hr.score_markers : hr.score_len;
            bool sw;
            if (sq == sr) sw = (!tie_by_rank && !Q->names.empty() && !R->names.empty()) ? Q->names[q] > R->names[r] : Q->rank[q] > R->rank[r];   // query_file_name > ref_file_name
            else sw = sq > sr;
            const skh_sketch_set::HalfLite& A = sw ? hr : hq; const skh_sketch_set::HalfLite& B = sw ? hq : hr;   // A: enumerated side (chain.rs:652-660)
            const uint32_t gb = sw ? q : r;
            job.recs[i] = PairRec{r, q, (sw ? 4u : 0u) | (empty ? 8u : 0u) | (rs << 8) | (qs << 20), 0u};
            job.hp[i] = HostPair{empty ? 0u : A.n_pos, B.nbk, 0u, sw ? 4u : 0u, A.nctg, B.nctg};
            if (stats || wide_run) {                                                   // (the rare paths take the full records)
            const skh_sketch_set::GenomeHalf& A = sw ? R->halves[r] : Q->halves[q]; const skh_sketch_set::GenomeHalf& B = sw ? Q->halves[q] : R->halves[r];
            if (stats) { job.host_go_a[i] = A.host_goff; job.host_go_b[i] = B.host_goff; }
            if (wide_run) {
                const bool aw = A.g64 != nullptr, bw = B.g64 != nullptr;
                WidePair& wp = job.wps[i];
                wp.a_g = aw ? CoArr{A.g64, 1u} : CoArr{A.g, 0u};
                wp.a_goff = aw ? CoArr{A.goff64, 1u} : CoArr{A.goff, 0u}; wp.b_goff = bw ? CoArr{B.goff64, 1u} : CoArr{B.goff, 0u};
                wp.b_g64 = B.g64; wp.a_is_index = aw ? 1u : 0u; wp.pad = 0;
                if (stats) {
                    job.host_go_a64[i] = aw ? CoArr{A.host_goff64, 1u} : CoArr{A.host_goff, 0u};
                    job.host_go_b64[i] = bw ? CoArr{B.host_goff64, 1u} : CoArr{B.host_goff, 0u};
                }
            }
            }
            job.pair_key[i] = gb + 3u * (sw ? n_rsets + qs : rs);                    // tiles probing the same sketch share an XCD
            job.chunk_bound[i] = A.chunk_bound;
        }
        tr.mark("host: pair descriptors");
        if (tables_pending && !pending_done) {
            // the tables these descriptors point into were being built meanwhile: wait for them; a genome that had to take another salt (common.h
            // table_hash; the descriptors carry salt 0) is rare -- then the salts are put in again
            pending_done = true; (*tables_pending)();
            // What the build fixed only now: the salts of crowded genomes, and -- a set made with SKH_SKETCH_COMPACT | SKH_SKETCH_DEFER_TABLES -- the place of the
            // list storage, which the build cuts to size and moves.  The cached halves were made before: they go, and the probed side of this batch's
            // descriptors is filled in again from fresh ones (the later batches of a split call are made from them anyway).
            for (uint32_t x = 0; x < n_rsets; x++) { drop_halves(Rsets[x]); genome_halves(Rsets[x]); }
            for (uint32_t x = 0; x < n_qsets; x++) { drop_halves(Qsets[x]); genome_halves(Qsets[x]); }
            device_tables();                                                          // (the pair records name genomes, not addresses: nothing else to put right)
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t p = idx ? idx[i] : i;
                const skh_sketch_set *R, *Q; uint32_t rs, qs; halves_of(p, R, Q, rs, qs);
                job.hp[i].b_nbk = ((job.hp[i].flags & 4u) ? Q->lite[pair_query[p]] : R->lite[pair_ref[p]]).nbk;   // B = the query when switched
            }
        }
        if (!split) { chain_run<Narrow>(ctx, job, out, stats); break; }
        std::vector<skh_ani_result> part(n); std::vector<skh_chain_stats> part_st(stats ? n : 0);
        if (wide_run) chain_run<Wide>(ctx, job, part.data(), stats ? part_st.data() : nullptr);
        else chain_run<Narrow>(ctx, job, part.data(), stats ? part_st.data() : nullptr);
        for (uint32_t i = 0; i < n; i++) { out[idx[i]] = part[i]; if (stats) stats[idx[i]] = part_st[i]; }
    }
}

}  // namespace skh
