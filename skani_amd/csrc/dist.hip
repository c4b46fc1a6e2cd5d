// dist.hip -- the all-vs-all triangle (triangle.rs:55-105) over several GPUs, one process per GPU, below the C ABI.
//
// The reference runs the triangle on one shared Vec<Sketch> with a work-stealing thread pool (nested par_iter, triangle.rs:71-90).  Here the
// genomes are spread over the ranks (each rank sketched its own), and what replaces the work stealing is a balanced, order-independent
// assignment of the screened pairs that every rank computes for itself from the same all-gathered candidate list:
//   * connected components of the candidate graph (clusters of related genomes) stay whole -- all pairs of a cluster are chained where its
//     sketches are, so every sketch travels at most once per cluster;
//   * a component too heavy for an even split is cut into (row-block x column-block) tiles of its genome list;
//   * units (components and tiles) are dealt out longest-first to the least loaded rank (cost of a pair = both genomes' marker counts, a proxy
//     for the two sketches the join has to read).
// Exchange steps (Transport: RCCL on device buffers, or caller-supplied host collectives): 3 small all-gathers of per-rank / per-genome
// tables, 1 all-gather of the marker sets (device), 2 of the candidate pairs, 1 all-to-all of the sketches that have to move (device:
// seed + padded-position arrays, 8 bytes per seed position), 2 all-gathers of the results.  No collective inside the pair pipeline.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <unordered_map>

#include "internal.h"

skh_comm::~skh_comm() { delete t; }

namespace skh {

namespace {

// ---- host-memory collectives supplied by the caller; device buffers are staged through host memory
struct HostTransport : Transport {
    skh_host_collectives hc;
    void all_gather(skh_ctx* ctx, const void* send, void* recv, size_t bytes, bool device) override {
        if (!device) { if (hc.all_gather(hc.user, send, recv, bytes)) throw Error("host all_gather failed"); return; }
        std::vector<char> hs(bytes ? bytes : 1), hr((size_t)world * bytes + 1);
        d2h(hs.data(), send, bytes, ctx->stream); dsync(ctx->stream);
        if (hc.all_gather(hc.user, hs.data(), hr.data(), bytes)) throw Error("host all_gather failed");
        h2d(recv, hr.data(), (size_t)world * bytes, ctx->stream); dsync(ctx->stream);
    }
    void all_to_all_v(skh_ctx* ctx, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                      const uint64_t* recv_off, bool device) override {
        if (!device) { if (hc.all_to_all_v(hc.user, send, send_cnt, send_off, recv, recv_cnt, recv_off)) throw Error("host all_to_all_v failed"); return; }
        uint64_t sb = 0, rb = 0;
        for (int r = 0; r < world; r++) { sb = std::max(sb, send_off[r] + send_cnt[r]); rb = std::max(rb, recv_off[r] + recv_cnt[r]); }
        std::vector<char> hs(sb + 1), hr(rb + 1);
        d2h(hs.data(), send, sb, ctx->stream); dsync(ctx->stream);
        if (hc.all_to_all_v(hc.user, hs.data(), send_cnt, send_off, hr.data(), recv_cnt, recv_off)) throw Error("host all_to_all_v failed");
        h2d(recv, hr.data(), rb, ctx->stream); dsync(ctx->stream);
    }
};

// copies n_seg segments of 32-bit words: segment s = src[seg[3s]] .. (seg[3s+2] words) -> dst[seg[3s+1]] ..; one workgroup per (segment, 4096-word slice)
__global__ __launch_bounds__(256) void copy_segments_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const uint64_t* __restrict__ seg, uint32_t n_seg) {
    const uint32_t s = blockIdx.x;
    if (s >= n_seg) return;
    const uint64_t so = seg[3 * (uint64_t)s], dof = seg[3 * (uint64_t)s + 1], n = seg[3 * (uint64_t)s + 2];
    for (uint64_t i = (uint64_t)blockIdx.y * 4096 + threadIdx.x; i < n; i += (uint64_t)gridDim.y * 4096) {
#pragma unroll
        for (uint32_t u = 0; u < 16; u++) { const uint64_t x = i + 256u * u; if (x < n && x < (i - threadIdx.x) + 4096) dst[dof + x] = src[so + x]; }
    }
}
void copy_segments(skh_ctx* ctx, const uint32_t* src, uint32_t* dst, const std::vector<uint64_t>& seg) {
    const uint32_t n_seg = (uint32_t)(seg.size() / 3);
    if (!n_seg) return;
    uint64_t mx = 0; for (uint32_t s = 0; s < n_seg; s++) mx = std::max(mx, seg[3 * (size_t)s + 2]);
    if (!mx) return;
    uint64_t* d_seg = ctx->arena.get<uint64_t>(seg.size());
    h2d(d_seg, seg.data(), seg.size() * 8, ctx->stream);
    const uint32_t slices = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (mx + 4095) / 4096), 64);
    SKH_LAUNCH(copy_segments_kernel, dim3(n_seg, slices), 256, 0, ctx->stream, src, dst, (const uint64_t*)d_seg, n_seg);
    check_launch("copy_segments");
}

struct Dsu {
    std::vector<uint32_t> p;
    explicit Dsu(uint32_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
    uint32_t find(uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(uint32_t a, uint32_t b) { a = find(a); b = find(b); if (a != b) { if (a < b) p[b] = a; else p[a] = b; } }   // root = smallest member: deterministic
};

}  // namespace

// The balanced assignment (identical on every rank: it only depends on the all-gathered inputs).  owner[p] = rank that chains pair p.
// holder (may be empty): the rank that holds each genome's sketch.
void assign_pairs(uint32_t n_genomes, const std::vector<uint32_t>& pi, const std::vector<uint32_t>& pj, const std::vector<uint64_t>& weight /* per genome */,
                  const std::vector<int>& holder, int world, std::vector<uint8_t>& owner, std::vector<uint64_t>& units_of, std::vector<uint64_t>& load) {
    const size_t NP = pi.size();
    owner.assign(NP, 0); load.assign(world, 0); units_of.assign(world, 0);
    if (!NP) return;
    Dsu dsu(n_genomes);
    for (size_t p = 0; p < NP; p++) dsu.unite(pi[p], pj[p]);
    std::vector<uint64_t> comp_cost(n_genomes, 0); uint64_t total = 0;
    auto pair_cost = [&](size_t p) { return std::max<uint64_t>(1, weight[pi[p]] + weight[pj[p]]); };
    for (size_t p = 0; p < NP; p++) { const uint64_t c = pair_cost(p); comp_cost[dsu.find(pi[p])] += c; total += c; }
    const uint64_t max_unit = std::max<uint64_t>(1, total / ((uint64_t)world * 16));
    // heavy components: position of every member inside its component (ascending global index) and the component's tile grid
    std::vector<uint32_t> pos_in(n_genomes, 0), comp_n(n_genomes, 0), gsize(n_genomes, 0);
    bool any_heavy = false;
    for (uint32_t g = 0; g < n_genomes; g++) { const uint32_t r = dsu.find(g); if (comp_cost[r] > max_unit) { pos_in[g] = comp_n[r]++; any_heavy = true; } }
    if (any_heavy)
        for (uint32_t r = 0; r < n_genomes; r++)
            if (comp_n[r]) {
                const double t = std::ceil(std::sqrt(2.0 * (double)comp_cost[r] / (double)max_unit));
                const uint32_t T = (uint32_t)std::min<double>(std::max(1.0, t), (double)comp_n[r]);
                gsize[r] = (comp_n[r] + T - 1) / T;
            }
    // units: a light component is one unit (looked up by its root); the tiles of a heavy one are keyed (root, row group, column group)
    std::vector<uint32_t> unit_of(NP);
    std::vector<int32_t> comp_unit(n_genomes, -1);
    std::unordered_map<uint64_t, uint32_t> tile_unit;
    std::vector<uint64_t> ukey, ucost; std::vector<uint32_t> aff;                   // aff[u * world + r]: pair end points of unit u whose sketch rank r holds
    auto new_unit = [&](uint64_t k) { ukey.push_back(k); ucost.push_back(0); if (!holder.empty()) aff.resize(aff.size() + world, 0); return (uint32_t)(ukey.size() - 1); };
    for (size_t p = 0; p < NP; p++) {
        const uint32_t r = dsu.find(pi[p]);
        uint32_t u;
        if (!gsize[r]) { if (comp_unit[r] < 0) comp_unit[r] = (int32_t)new_unit(((uint64_t)r << 32) | 0xFFFFFFFFull); u = (uint32_t)comp_unit[r]; }
        else {
            uint32_t a = pos_in[pi[p]] / gsize[r], b = pos_in[pj[p]] / gsize[r];
            if (a > b) std::swap(a, b);
            const uint64_t k = ((uint64_t)r << 32) | ((uint64_t)std::min(a, 0xFFFEu) << 16) | std::min(b, 0xFFFEu);
            auto it = tile_unit.find(k);
            if (it == tile_unit.end()) { u = new_unit(k); tile_unit.emplace(k, u); } else u = it->second;
        }
        unit_of[p] = u; ucost[u] += pair_cost(p);
        if (!holder.empty()) { aff[(size_t)u * world + holder[pi[p]]]++; aff[(size_t)u * world + holder[pj[p]]]++; }
    }
    const uint32_t NU = (uint32_t)ukey.size();
    std::vector<uint32_t> order(NU); std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ucost[a] != ucost[b] ? ucost[a] > ucost[b] : ukey[a] < ukey[b]; });
    std::vector<uint8_t> unit_rank(NU, 0); std::vector<uint8_t> placed(NU, 0);
    const uint64_t target = (total + (uint64_t)world - 1) / (uint64_t)world;
    // pass 1, "stay home unless home is full": a unit goes to the rank that already holds most of its sketches while that rank stays within
    // an even share (longest first), so clusters that live on one rank stay there and nothing travels
    if (!holder.empty())
        for (uint32_t u : order) {
            int home = 0;
            for (int r = 1; r < world; r++) if (aff[(size_t)u * world + r] > aff[(size_t)u * world + home]) home = r;
            if (load[home] + ucost[u] <= target) { load[home] += ucost[u]; units_of[home]++; unit_rank[u] = (uint8_t)home; placed[u] = 1; }
        }
    // pass 2: what is left, longest first, to the least loaded rank (ties: the rank holding more of the unit's sketches, then the lowest)
    for (uint32_t u : order) {
        if (placed[u]) continue;
        int best = 0;
        for (int r = 1; r < world; r++) {
            if (load[r] < load[best]) best = r;
            else if (load[r] == load[best] && !holder.empty() && aff[(size_t)u * world + r] > aff[(size_t)u * world + best]) best = r;
        }
        load[best] += ucost[u]; units_of[best]++; unit_rank[u] = (uint8_t)best;
    }
    for (size_t p = 0; p < NP; p++) owner[p] = unit_rank[unit_of[p]];
}

void triangle_distributed(skh_ctx* ctx, Transport& T, const skh_sketch_set* L, double identity, int rescue_small, const skh_map_params& mp,
                          std::vector<uint32_t>& out_i, std::vector<uint32_t>& out_j, std::vector<skh_ani_result>& out_res, uint64_t* n_chained, skh_dist_stats* stats) {
    const int W = T.world, me = T.rank;
    if (W < 1 || me < 0 || me >= W || W > 255) throw std::invalid_argument("bad communicator");
    if (L->ctx != ctx) throw std::invalid_argument("the local sketch set belongs to another context");
    out_i.clear(); out_j.clear(); out_res.clear();
    skh_dist_stats st{};
    StageTrace tr(ctx);
    // exchange steps are timed on the host clock: the collectives may run on the transport's own stream
    std::chrono::steady_clock::time_point ex_t0; double exch_ms = 0;
    auto ex_begin = [&] { dsync(ctx->stream); ex_t0 = std::chrono::steady_clock::now(); };
    auto ex_end = [&] { dsync(ctx->stream); exch_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ex_t0).count(); };
    // Failures that only ONE rank can see (out of memory, a table overflow, a failed launch) must not leave the others waiting in the next
    // collective: every local phase runs under `local`, which keeps the first error; `agree` -- one small all-gather of status words, placed in
    // front of the collective that follows the phase -- makes every rank throw together, naming the rank that failed.  (Errors every rank derives
    // from the same all-gathered data, like the parameter check below, need no agreement.)
    std::string local_err;
    uint32_t local_phase = 0;                                                       // (SKH_TUNE_DIST_FAIL = n makes the n-th local phase of this rank fail: the tests' fault injection)
    auto local = [&](auto&& body) {
        local_phase++;
        if (!local_err.empty()) return;
        try { if (ctx->tune.dist_fail == local_phase) throw Error("injected failure (SKH_TUNE_DIST_FAIL)"); body(); }
        catch (const std::exception& e) { local_err = e.what(); }
        catch (...) { local_err = "unknown error"; }
    };
    auto stop_together = [&](const char* phase, int r) {                           // rank r reported a failure: every rank throws
        device_sync_all();                                                          // nothing queued may outlive the buffers the unwinding frees
        if (r == me) throw Error(std::string("distributed triangle, ") + phase + ": " + local_err);
        throw Error(std::string("distributed triangle, ") + phase + ": rank " + std::to_string(r) + " failed (its own error message says why); all ranks stop");
    };
    // a count every rank contributes anyway + its status in ONE small all-gather (where a phase is followed by such a gather the agreement is free)
    auto gather_count_and_status = [&](uint64_t my_count, std::vector<uint64_t>& counts, const char* phase) {
        uint64_t mine2[2] = {my_count, local_err.empty() ? 0ull : 1ull}; std::vector<uint64_t> all((size_t)W * 2);
        T.all_gather(ctx, mine2, all.data(), sizeof(mine2), false);
        counts.resize(W);
        for (int r = 0; r < W; r++) counts[r] = all[(size_t)r * 2];
        for (int r = 0; r < W; r++) if (all[(size_t)r * 2 + 1]) stop_together(phase, r);
    };
    auto agree = [&](const char* phase) {
        uint64_t mine_ok = local_err.empty() ? 0 : 1; std::vector<uint64_t> all(W);
        T.all_gather(ctx, &mine_ok, all.data(), 8, false);
        for (int r = 0; r < W; r++)
            if (all[r]) {
                device_sync_all();                                                  // nothing queued may outlive the buffers the unwinding frees
                if (r == me) throw Error(std::string("distributed triangle, ") + phase + ": " + local_err);
                throw Error(std::string("distributed triangle, ") + phase + ": rank " + std::to_string(r) + " failed (its own error message says why); all ranks stop");
            }
    };
    // ---- 1. who holds what: per rank (genomes, seed positions, markers, contigs, c, k, marker_c, seeding mode), then per genome, then the contig lengths
    const uint32_t nL = L->n_genomes;
    uint64_t mine[8] = {nL, L->pos_off[nL], L->mk_off[nL], L->ctg_off[nL], L->params.c, L->params.k, L->params.marker_c, (uint64_t)L->params.seeding_mode | (L->wide ? 256u : 0u)};
    std::vector<uint64_t> cnt((size_t)W * 8);
    ex_begin();
    T.all_gather(ctx, mine, cnt.data(), sizeof(mine), false);
    std::vector<uint64_t> base(W + 1, 0);
    uint64_t max_n = 0, max_m = 0, max_c = 0;
    // A rank with a wide set (a genome beyond 31-bit padded coordinates, internal.h): position records are not portable then (a wide genome's are
    // indices beside 64-bit coordinates), so ALL ranks exchange (position in contig, contig << 1 | canonical) -- the C ABI's form, 8 bytes instead
    // of 4 -- and the chained set is made through the import path, which decides per genome from the contig lengths, the same on every rank.
    bool wide_any = false;
    for (int r = 0; r < W; r++) {
        if (cnt[r * 8 + 4] != mine[4] || cnt[r * 8 + 5] != mine[5] || cnt[r * 8 + 6] != mine[6] || (cnt[r * 8 + 7] & 255u) != (mine[7] & 255u))
            throw std::invalid_argument("the ranks sketched with different c / k / marker_c / seeding mode");
        if (cnt[r * 8 + 7] & 256u) wide_any = true;                                 // (every rank sees the same table)
        base[r + 1] = base[r] + cnt[r * 8]; max_n = std::max(max_n, cnt[r * 8]); max_m = std::max(max_m, cnt[r * 8 + 2]); max_c = std::max(max_c, cnt[r * 8 + 3]);
    }
    const uint64_t N64 = base[W];
    if (N64 >= (1ull << 21)) throw std::invalid_argument("more than 2M genomes in one distributed triangle");
    const uint32_t N = (uint32_t)N64;
    st.n_genomes_total = N;
    constexpr uint32_t GF = 5;                                                      // per genome: seed positions, markers, contigs, total length, rank
    // (one more word behind the per-genome fields: this rank's status after preparing its marker buffers below -- the agreement rides on this gather)
    std::vector<uint64_t> gm_mine((size_t)std::max<uint64_t>(max_n, 1) * GF + 1, 0), gm_all((size_t)W * gm_mine.size());
    for (uint32_t g = 0; g < nL; g++) {
        gm_mine[g * GF + 0] = L->pos_off[g + 1] - L->pos_off[g]; gm_mine[g * GF + 1] = L->mk_off[g + 1] - L->mk_off[g];
        gm_mine[g * GF + 2] = L->ctg_off[g + 1] - L->ctg_off[g]; gm_mine[g * GF + 3] = L->total_len[g]; gm_mine[g * GF + 4] = L->rank[g];
    }
    // this rank's marker set, padded, in the staging buffers of the marker all-gather further down; how that went travels with the table
    const uint64_t pad = std::max<uint64_t>(max_m, 1);
    uint64_t *d_send = nullptr, *d_recv = nullptr;
    local([&] {                                                                     // (local phase 1)
        d_send = ctx->arena.get<uint64_t>(pad); d_recv = ctx->arena.get<uint64_t>(pad * W);
        if (mine[2]) d2d(d_send, L->markers.p, mine[2] * 8, ctx->stream);
        if (pad > mine[2]) dzero(d_send + mine[2], (pad - mine[2]) * 8, ctx->stream);
        dsync(ctx->stream);
    });
    gm_mine.back() = local_err.empty() ? 0 : 1;
    T.all_gather(ctx, gm_mine.data(), gm_all.data(), gm_mine.size() * 8, false);
    for (int r = 0; r < W; r++) if (gm_all[(size_t)(r + 1) * gm_mine.size() - 1]) stop_together("marker buffers", r);
    std::vector<uint32_t> cl_mine(std::max<uint64_t>(max_c, 1), 0), cl_all((size_t)W * cl_mine.size());
    std::copy(L->ctg_len.begin(), L->ctg_len.end(), cl_mine.begin());
    T.all_gather(ctx, cl_mine.data(), cl_all.data(), cl_mine.size() * 4, false);
    auto G = [&](uint32_t g, uint32_t f) {                                          // field f of global genome g
        const int r = (int)(std::upper_bound(base.begin(), base.end(), (uint64_t)g) - base.begin()) - 1;
        return gm_all[(size_t)r * gm_mine.size() + (size_t)(g - base[r]) * GF + f];
    };
    std::vector<int> rank_of(N); std::vector<uint64_t> g_npos(N), g_nmk(N), g_nctg(N), g_len(N), g_rank(N), g_ctg0(N);   // g_ctg0: first contig in cl_all
    {
        uint32_t g = 0;
        for (int r = 0; r < W; r++) {
            uint64_t c0 = (uint64_t)r * cl_mine.size();
            for (uint64_t x = 0; x < cnt[r * 8]; x++, g++) {
                rank_of[g] = r; g_npos[g] = G(g, 0); g_nmk[g] = G(g, 1); g_nctg[g] = G(g, 2); g_len[g] = G(g, 3); g_rank[g] = G(g, 4); g_ctg0[g] = c0; c0 += g_nctg[g];
            }
        }
    }
    // ---- 2. all marker sets, on every rank (device memory), as a markers-only sketch set in global genome order
    skh_sketch_set S; S.ctx = ctx; S.params = L->params; S.n_genomes = N;
    S.mk_off.assign(N + 1, 0); for (uint32_t g = 0; g < N; g++) S.mk_off[g + 1] = S.mk_off[g] + g_nmk[g];
    const uint64_t MT = S.mk_off[N];
    {
        T.all_gather(ctx, d_send, d_recv, pad * 8, true);
        local([&] {                                                                 // (local phase 2; agreed on with the candidate counts)
            S.markers.alloc(MT ? MT : 1);
            for (int r = 0; r < W; r++) if (cnt[r * 8 + 2]) d2d(S.markers.p + S.mk_off[base[r]], d_recv + (uint64_t)r * pad, cnt[r * 8 + 2] * 8, ctx->stream);
            S.d_mk_off.alloc(N + 1); h2d(S.d_mk_off.p, S.mk_off.data(), (N + 1) * 8, ctx->stream);
            dsync(ctx->stream);
        });
    }
    ex_end();
    ctx->arena.reset();
    tr.mark("dist: tables + markers gathered");
    // ---- 3. screen: this rank's rows of the triangle; rows are cut so that every rank gets the same number of cells (row i has N - 1 - i)
    std::vector<uint32_t> rb(W + 1, 0);
    {
        const double cells = (double)N * (double)(N > 0 ? N - 1 : 0) / 2.0;
        for (int r = 1; r < W; r++) {                                               // smallest x with x (2N - 1 - x) / 2 >= r / W * cells
            const double target = cells * r / W, b = 2.0 * N - 1.0;
            double x = (b - std::sqrt(std::max(0.0, b * b - 8.0 * target))) / 2.0;
            rb[r] = (uint32_t)std::min<double>(std::max(std::ceil(x), (double)rb[r - 1]), (double)N);
        }
        rb[W] = N;
    }
    st.screen_row_begin = rb[me]; st.screen_row_end = rb[me + 1];
    std::vector<uint32_t> my_i, my_j;
    local([&] {                                                                     // (local phase 3)
        if (rb[me + 1] <= rb[me]) return;
        Stopwatch sw(ctx, &ctx->timings.screen_ms);
        screen_pairs(ctx, &S, nullptr, identity, SKH_SCREEN_REFS, rescue_small, my_i, my_j, rb[me], rb[me + 1]);
    });
    ctx->arena.reset();
    tr.mark("dist: screen rows");
    // ---- 4. the candidate list, everywhere (host memory; sorted by (i, j) because the row blocks ascend with the rank)
    ex_begin();
    uint64_t my_np = my_i.size(); std::vector<uint64_t> np_all(W);
    gather_count_and_status(my_np, np_all, "marker sets / screen");
    uint64_t max_np = 1, NP64 = 0; for (int r = 0; r < W; r++) { max_np = std::max(max_np, np_all[r]); NP64 += np_all[r]; }
    std::vector<uint32_t> pi, pj;
    {
        std::vector<uint32_t> sendp(max_np * 2, 0), recvp((size_t)W * max_np * 2);
        std::copy(my_i.begin(), my_i.end(), sendp.begin()); std::copy(my_j.begin(), my_j.end(), sendp.begin() + max_np);
        T.all_gather(ctx, sendp.data(), recvp.data(), sendp.size() * 4, false);
        pi.reserve(NP64); pj.reserve(NP64);
        for (int r = 0; r < W; r++) {
            const uint32_t* b = recvp.data() + (size_t)r * max_np * 2;
            pi.insert(pi.end(), b, b + np_all[r]); pj.insert(pj.end(), b + max_np, b + max_np + np_all[r]);
        }
    }
    ex_end();
    const size_t NP = pi.size();
    st.n_candidate_pairs_total = NP;
    if (n_chained) *n_chained = NP;
    // ---- 5. the assignment
    std::vector<uint8_t> owner; std::vector<uint64_t> load, units_of;
    assign_pairs(N, pi, pj, g_nmk, rank_of, W, owner, units_of, load);
    st.cost_mine = load[me]; st.n_units_mine = units_of[me];
    for (int r = 0; r < W; r++) { st.cost_total += load[r]; st.n_units_total += units_of[r]; }
    tr.mark("dist: pairs gathered + assigned");
    // ---- 6. which sketches move: mark[g][r] = rank r chains a pair with genome g
    std::vector<std::vector<uint32_t>> send_to(W), recv_from(W);                   // global ids, ascending
    std::vector<uint32_t> wk_ids;                                                   // the genomes THIS rank chains (its own that stay + the ones it receives), ascending
    {
        std::vector<uint8_t> mark((size_t)N * W, 0);
        for (size_t p = 0; p < NP; p++) { mark[(size_t)pi[p] * W + owner[p]] = 1; mark[(size_t)pj[p] * W + owner[p]] = 1; }
        for (uint32_t g = 0; g < N; g++) {
            if (mark[(size_t)g * W + me]) wk_ids.push_back(g);
            for (int r = 0; r < W; r++)
                if (mark[(size_t)g * W + r] && rank_of[g] != r) {
                    if (rank_of[g] == me) send_to[r].push_back(g);
                    if (r == me) recv_from[rank_of[g]].push_back(g);
                }
        }
    }
    uint32_t nR = 0; for (int r = 0; r < W; r++) nR += (uint32_t)recv_from[r].size();
    st.n_genomes_received = nR;
    // The work set: when sketches arrive, ONE set is made of the genomes this rank chains -- the local ones copied, the received ones from the
    // exchange buffer, in ascending global index -- and its seed tables are built once.  A local sketch whose clusters went to other ranks is never
    // indexed here (with deferred tables: skh_sketch_genomes_ex), a sketch that travels is indexed only where it arrives.  Nothing received: the local
    // set itself is chained.
    std::unique_ptr<skh_sketch_set> Wk;
    std::vector<uint32_t> wk_index(N, 0xFFFFFFFFu);                                 // global genome -> index in the chained set
    ex_begin();
    {
        // send buffer per destination: [seeds of all its genomes][padded positions of all its genomes]  (32-bit words; with a wide set somewhere:
        // [seeds][positions in contig][contig << 1 | canonical])
        const uint64_t NF = wide_any ? 3 : 2;                                       // 32-bit fields per seed position
        std::vector<uint64_t> s_cnt(W, 0), s_off(W, 0), r_cnt(W, 0), r_off(W, 0), seg_s, seg_g, seg_c;
        uint64_t sw = 0;
        for (int r = 0; r < W; r++) {
            uint64_t words = 0; for (uint32_t g : send_to[r]) words += g_npos[g];
            s_off[r] = sw * 4; s_cnt[r] = words * NF * 4;
            uint64_t at = sw;
            for (uint32_t g : send_to[r]) {
                const uint64_t lp = L->pos_off[g - base[me]];
                seg_s.insert(seg_s.end(), {lp, at, g_npos[g]}); seg_g.insert(seg_g.end(), {lp, at + words, g_npos[g]});   // the position parts follow the seed parts
                if (wide_any) seg_c.insert(seg_c.end(), {lp, at + 2 * words, g_npos[g]});
                at += g_npos[g];
            }
            sw += words * NF;
        }
        uint64_t rw = 0; std::vector<uint64_t> r_words(W, 0);
        std::vector<uint64_t> recv_at(N, 0);                                        // word offset of a received genome's seeds inside its source's block
        for (int r = 0; r < W; r++) {
            for (uint32_t g : recv_from[r]) { recv_at[g] = r_words[r]; r_words[r] += g_npos[g]; }
            r_off[r] = rw * 4; r_cnt[r] = r_words[r] * NF * 4; rw += r_words[r] * NF;
        }
        st.bytes_sent = sw * 4; st.bytes_received = rw * 4;
        uint32_t *d_send = nullptr, *d_recv = nullptr, *l_pos = nullptr, *l_cc = nullptr;   // l_pos / l_cc: the local set's positions in the C ABI's form (wide_any)
        local([&] {
            d_send = ctx->arena.get<uint32_t>(sw + 1); d_recv = ctx->arena.get<uint32_t>(rw + 1);
            copy_segments(ctx, L->p_seed.p, d_send, seg_s);
            if (wide_any) {
                const uint64_t PL = L->pos_off[nL];
                l_pos = ctx->arena.get<uint32_t>(PL + 1); l_cc = ctx->arena.get<uint32_t>(PL + 1);
                unpack_positions(ctx, L, 0, PL, l_pos, l_cc);
                copy_segments(ctx, l_pos, d_send, seg_g); copy_segments(ctx, l_cc, d_send, seg_c);
            } else copy_segments(ctx, L->p_g.p, d_send, seg_g);
            dsync(ctx->stream);
        });
        agree("sketch exchange buffers");
        T.all_to_all_v(ctx, d_send, s_cnt.data(), s_off.data(), d_recv, r_cnt.data(), r_off.data(), true);
        local([&] {                                                                 // (local phase 5)
            if (!nR) { for (uint32_t g : wk_ids) wk_index[g] = (uint32_t)(g - base[me]); return; }
            const uint32_t nW = (uint32_t)wk_ids.size();
            Wk.reset(new skh_sketch_set());
            Wk->ctx = ctx; Wk->params = L->params; Wk->n_genomes = nW;
            Wk->rank.resize(nW); Wk->pos_off.assign(nW + 1, 0); Wk->mk_off.assign(nW + 1, 0); Wk->ctg_off.assign(nW + 1, 0); Wk->total_len.resize(nW);
            for (uint32_t x = 0; x < nW; x++) {
                const uint32_t g = wk_ids[x];
                wk_index[g] = x;
                Wk->rank[x] = (uint32_t)g_rank[g]; Wk->total_len[x] = g_len[g];
                Wk->pos_off[x + 1] = Wk->pos_off[x] + g_npos[g]; Wk->mk_off[x + 1] = Wk->mk_off[x] + g_nmk[g]; Wk->ctg_off[x + 1] = Wk->ctg_off[x] + g_nctg[g];
                for (uint64_t c = 0; c < g_nctg[g]; c++) Wk->ctg_len.push_back(cl_all[g_ctg0[g] + c]);
            }
            finalize_metadata(Wk.get());
            const uint64_t PW = Wk->pos_off[nW], MW = Wk->mk_off[nW];
            Wk->p_seed.alloc(PW ? PW : 1); Wk->markers.alloc(MW ? MW : 1);
            uint32_t *w_pos = nullptr, *w_cc = nullptr;                             // wide_any: the chained set's positions in the C ABI's form, for the import path
            if (wide_any) { w_pos = ctx->arena.get<uint32_t>(PW + 1); w_cc = ctx->arena.get<uint32_t>(PW + 1); } else Wk->p_g.alloc(PW);
            std::vector<uint64_t> ls, lg, rs, rg, rc, mseg;                         // local / received seed and position segments; markers: already here (step 2), 64-bit = two words
            for (uint32_t x = 0; x < nW; x++) {
                const uint32_t g = wk_ids[x]; const uint64_t n = g_npos[g], dst = Wk->pos_off[x];
                if (rank_of[g] == me) { const uint64_t lp = L->pos_off[g - base[me]]; ls.insert(ls.end(), {lp, dst, n}); lg.insert(lg.end(), {lp, dst, n}); }
                else {
                    const uint64_t b0 = r_off[rank_of[g]] / 4, rwd = r_words[rank_of[g]];
                    rs.insert(rs.end(), {b0 + recv_at[g], dst, n}); rg.insert(rg.end(), {b0 + rwd + recv_at[g], dst, n});
                    if (wide_any) rc.insert(rc.end(), {b0 + 2 * rwd + recv_at[g], dst, n});
                }
                mseg.insert(mseg.end(), {S.mk_off[g] * 2, Wk->mk_off[x] * 2, g_nmk[g] * 2});
            }
            copy_segments(ctx, L->p_seed.p, Wk->p_seed.p, ls); copy_segments(ctx, d_recv, Wk->p_seed.p, rs);
            if (wide_any) {
                copy_segments(ctx, l_pos, w_pos, lg); copy_segments(ctx, l_cc, w_cc, lg);
                copy_segments(ctx, d_recv, w_pos, rg); copy_segments(ctx, d_recv, w_cc, rc);
            } else { copy_segments(ctx, L->p_g.p, Wk->p_g.p, lg); copy_segments(ctx, d_recv, Wk->p_g.p, rg); }
            copy_segments(ctx, (const uint32_t*)S.markers.p, (uint32_t*)Wk->markers.p, mseg);
            Wk->d_mk_off.alloc(nW + 1); h2d(Wk->d_mk_off.p, Wk->mk_off.data(), (nW + 1) * 8, ctx->stream);
            if (wide_any) { Stopwatch sw2(ctx, &ctx->timings.sketch_build_ms); build_sketch_tables(ctx, Wk.get(), w_pos, w_cc); }   // (the position arrays live in the arena: the tables are made here)
            dsync(ctx->stream);
        });
    }
    ex_end();
    ctx->arena.reset();
    tr.mark("dist: sketches exchanged");
    const skh_sketch_set* CS = Wk ? Wk.get() : L;                                   // the set that is chained
    local([&] { if (wk_ids.empty()) return; Stopwatch sw(ctx, &ctx->timings.sketch_build_ms); ensure_tables(ctx, CS); });   // (local phase 6)
    ctx->arena.reset();
    tr.mark("dist: seed tables of the chained set");
    // ---- 7. chain this rank's pairs: ref = genome i, query = genome j (triangle.rs:89-98)
    std::vector<uint32_t> c_i, c_j, c_r, c_q;
    for (size_t p = 0; p < NP; p++) {
        if (owner[p] != me) continue;
        c_i.push_back(pi[p]); c_j.push_back(pj[p]); c_r.push_back(wk_index[pi[p]]); c_q.push_back(wk_index[pj[p]]);
    }
    st.n_pairs_mine = c_i.size();
    std::vector<skh_ani_result> res(c_i.size());
    // ties of switch_qr go by genome_rank on every rank: the work set carries no file names, and which rank chains a pair must not decide its orientation
    local([&] {                                                                     // (local phase 7)
        if (c_i.empty()) return;
        Stopwatch sw(ctx, &ctx->timings.chain_ms);
        chain_pairs(ctx, &CS, 1, nullptr, &CS, 1, nullptr, c_r.data(), c_q.data(), c_i.size(), mp, res.data(), nullptr, true);
    });
    ctx->arena.reset();
    tr.mark("dist: chain");
    // ---- 8. results (ani > 0.1, triangle.rs:99) gathered on every rank, sorted by (i, j)
    struct Row { uint32_t i, j; skh_ani_result r; };
    std::vector<Row> rows;
    for (size_t p = 0; p < res.size(); p++) if (res[p].ani > 0.1f) rows.push_back(Row{c_i[p], c_j[p], res[p]});
    ex_begin();
    uint64_t my_rows = rows.size(); std::vector<uint64_t> rows_all(W);
    gather_count_and_status(my_rows, rows_all, "seed tables / chaining");
    uint64_t max_rows = 1, tot_rows = 0; for (int r = 0; r < W; r++) { max_rows = std::max(max_rows, rows_all[r]); tot_rows += rows_all[r]; }
    std::vector<Row> sendr(max_rows), recvr((size_t)W * max_rows);
    memset((void*)sendr.data(), 0, sendr.size() * sizeof(Row));
    std::copy(rows.begin(), rows.end(), sendr.begin());
    T.all_gather(ctx, sendr.data(), recvr.data(), max_rows * sizeof(Row), false);
    ex_end();
    std::vector<Row> all; all.reserve(tot_rows);
    for (int r = 0; r < W; r++) all.insert(all.end(), recvr.begin() + (size_t)r * max_rows, recvr.begin() + (size_t)r * max_rows + rows_all[r]);
    std::sort(all.begin(), all.end(), [](const Row& a, const Row& b) { return a.i != b.i ? a.i < b.i : a.j < b.j; });
    out_i.resize(all.size()); out_j.resize(all.size()); out_res.resize(all.size());
    for (size_t x = 0; x < all.size(); x++) { out_i[x] = all[x].i; out_j[x] = all[x].j; out_res[x] = all[x].r; }
    ctx->timings.exchange_ms += (float)exch_ms;
    if (stats) *stats = st;
    tr.mark("dist: results gathered");
}

Transport* make_host_transport(const skh_host_collectives* hc, int rank, int world) {
    if (!hc || !hc->all_gather || !hc->all_to_all_v) throw std::invalid_argument("null collective");
    HostTransport* t = new HostTransport(); t->hc = *hc; t->rank = rank; t->world = world;
    return t;
}

}  // namespace skh
