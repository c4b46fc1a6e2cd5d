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
// Exchange steps (Transport: RCCL on device buffers, or caller-supplied host collectives), round 5:
//   1. ONE all-gather of per-rank tables (genomes, sizes, contig lengths, and per genome how many of its markers fall into each rank's part of the key range);
//   2. an 8-byte agreement, then ONE all-to-all of marker sets by key range (every rank receives its own part of every genome's sorted set: 1/W of the bytes an
//      all-gather would move; the row form of very large collections still all-gathers);
//   3. ONE all-gather of the non-zero cells of every rank's partial count matrix (device buffers the communicator keeps); every rank adds them up row by row in LDS
//      and applies the rule itself -- no candidate list travels;
//   4. an 8-byte agreement, then the asynchronous all-to-all of exactly the sketches that have to move (seed + padded-position arrays, 8 bytes per seed position),
//      hidden behind the home set's table build and the home pairs;
//   5. the result rows: a 16-byte gather of counts + status, then the rows to rank 0 (SKH_DIST_ROWS_TO_ROOT) or an all-gather (every rank returns the triangle).
// No collective inside the pair pipeline.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <thread>
#include <unordered_map>

#include "internal.h"

skh_comm::~skh_comm() { delete t; }

namespace skh {

namespace {

// ---- host-memory collectives supplied by the caller; device buffers are staged through host memory
struct HostTransport : Transport {
    skh_host_collectives hc;
    void all_gather(skh_ctx* ctx, const void* send, void* recv, size_t bytes, bool device) override {
        if (!device) { if (hc.all_gather(hc.user, send, recv, bytes)) throw PeerError("host all_gather failed"); return; }
        std::vector<char> hs(bytes ? bytes : 1), hr((size_t)world * bytes + 1);
        d2h(hs.data(), send, bytes, ctx->stream); dsync(ctx->stream);
        if (hc.all_gather(hc.user, hs.data(), hr.data(), bytes)) throw PeerError("host all_gather failed");
        h2d(recv, hr.data(), (size_t)world * bytes, ctx->stream); dsync(ctx->stream);
    }
    void all_to_all_v(skh_ctx* ctx, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                      const uint64_t* recv_off, bool device) override {
        if (!device) { if (hc.all_to_all_v(hc.user, send, send_cnt, send_off, recv, recv_cnt, recv_off)) throw PeerError("host all_to_all_v failed"); return; }
        uint64_t sb = 0, rb = 0;
        for (int r = 0; r < world; r++) { sb = std::max(sb, send_off[r] + send_cnt[r]); rb = std::max(rb, recv_off[r] + recv_cnt[r]); }
        std::vector<char> hs(sb + 1), hr(rb + 1);
        d2h(hs.data(), send, sb, ctx->stream); dsync(ctx->stream);
        if (hc.all_to_all_v(hc.user, hs.data(), send_cnt, send_off, hr.data(), recv_cnt, recv_off)) throw PeerError("host all_to_all_v failed");
        h2d(recv, hr.data(), rb, ctx->stream); dsync(ctx->stream);
    }
    // asynchronous form: the send buffer comes to the host at once, the caller's collective runs on a thread of its own (the calling thread is inside the
    // library, not inside the caller's runtime, and makes no other collective until _end), the received bytes go up in _end
    std::thread worker; std::vector<char> a_send, a_recv; int a_rc = 0; void* a_dst = nullptr; uint64_t a_rb = 0;
    std::chrono::steady_clock::time_point a_t0, a_t1; uint64_t a_wait_us = 0;
    ~HostTransport() override { if (worker.joinable()) worker.join(); }
    void exchange_begin(skh_ctx* ctx, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt,
                        const uint64_t* recv_off) override {
        if (worker.joinable()) throw Error("exchange_begin: an exchange is already open");
        uint64_t sb = 0, rb = 0;
        for (int r = 0; r < world; r++) { sb = std::max(sb, send_off[r] + send_cnt[r]); rb = std::max(rb, recv_off[r] + recv_cnt[r]); }
        a_send.assign(sb + 1, 0); a_recv.assign(rb + 1, 0); a_dst = recv; a_rb = rb; a_rc = 0; a_wait_us = 0;
        d2h(a_send.data(), send, sb, ctx->stream); dsync(ctx->stream);
        a_t0 = a_t1 = std::chrono::steady_clock::now();
        worker = std::thread([this, send_cnt, send_off, recv_cnt, recv_off] {
            a_rc = hc.all_to_all_v(hc.user, a_send.data(), send_cnt, send_off, a_recv.data(), recv_cnt, recv_off);
            a_t1 = std::chrono::steady_clock::now();
        });
    }
    void exchange_end(skh_ctx* ctx) override {
        if (!worker.joinable()) return;
        const auto w0 = std::chrono::steady_clock::now();
        worker.join();
        a_wait_us = (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        if (a_rc) throw PeerError("host all_to_all_v failed");
        h2d(a_dst, a_recv.data(), a_rb, ctx->stream); dsync(ctx->stream);          // (a pageable source: through the pinned ring, dev.h)
    }
    void exchange_times(uint64_t* total_us, uint64_t* wait_us) override {
        if (total_us) *total_us = (uint64_t)std::chrono::duration<double, std::micro>(a_t1 - a_t0).count();
        if (wait_us) *wait_us = a_wait_us;
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
// the two head words of every block of a gathered buffer, side by side
__global__ __launch_bounds__(256) void block_heads_kernel(const uint64_t* blocks, uint64_t block_words, uint32_t n_blocks, uint64_t* heads) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 2 * n_blocks) heads[t] = blocks[(uint64_t)(t >> 1) * block_words + (t & 1u)];
}
}  // namespace
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
namespace {

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
    std::vector<uint32_t> root(n_genomes);                                          // (flattened once: every later look-up is one load -- the plan runs on every rank's critical path)
    for (uint32_t g = 0; g < n_genomes; g++) root[g] = dsu.find(g);
    std::vector<uint64_t> comp_cost(n_genomes, 0); uint64_t total = 0;
    auto pair_cost = [&](size_t p) { return std::max<uint64_t>(1, weight[pi[p]] + weight[pj[p]]); };
    for (size_t p = 0; p < NP; p++) { const uint64_t c = pair_cost(p); comp_cost[root[pi[p]]] += c; total += c; }
    const uint64_t max_unit = std::max<uint64_t>(1, total / ((uint64_t)world * 16));
    // heavy components: position of every member inside its component (ascending global index) and the component's tile grid
    std::vector<uint32_t> pos_in(n_genomes, 0), comp_n(n_genomes, 0), gsize(n_genomes, 0);
    bool any_heavy = false;
    for (uint32_t g = 0; g < n_genomes; g++) { const uint32_t r = root[g]; if (comp_cost[r] > max_unit) { pos_in[g] = comp_n[r]++; any_heavy = true; } }
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
        const uint32_t r = root[pi[p]];
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

// One all-gather instead of "counts, then payload": every rank sends a 16-byte head (records, status) and its records in a block sized by the capacity the
// communicator remembers from its previous call (Transport::cap_*).  A collection is usually run again with the same sizes (bench.py: every step), so the
// first round fits; when some rank has more records than the capacity -- every rank reads that from the same gathered heads -- all ranks raise the
// capacity to the largest count and go round once more.  Returns the records of all ranks back to back (rank order) and the per-rank counts; a rank whose
// status word is set stops every rank (on_failed).
template <class Rec> struct Gathered { std::vector<const Rec*> block; std::vector<uint64_t> counts; uint64_t total = 0; };   // rank r's records: block[r][0 .. counts[r])
template <class Rec, class Failed>
static Gathered<Rec> gather_records(skh_ctx* ctx, Transport& T, const std::vector<Rec>& mine, bool my_status, uint64_t& cap, Failed&& on_failed) {
    static_assert(sizeof(Rec) % 8 == 0, "records keep the block 8-byte aligned");
    const int W = T.world;
    Gathered<Rec> G; G.block.assign(W, nullptr); G.counts.assign(W, 0);
    for (int round = 0;; round++) {
        const size_t block = 16 + (size_t)cap * sizeof(Rec);
        // the communicator's own staging (grown, never shrunk, not cleared: only a block's head and its first `count` records mean anything) -- fresh vectors of
        // config 4's result rows were 7 MB of page faults and zeroes on every rank and call
        if (T.g_send.size() < block) T.g_send.resize(block);
        if (T.g_recv.size() < (size_t)W * block) T.g_recv.resize((size_t)W * block);
        char* send = T.g_send.data(); char* recv = T.g_recv.data();
        uint64_t head[2] = {mine.size(), my_status ? 1ull : 0ull};
        memcpy(send, head, 16);
        if (mine.size() <= cap && !mine.empty()) memcpy(send + 16, mine.data(), mine.size() * sizeof(Rec));
        T.all_gather(ctx, send, recv, block, false);
        uint64_t mx = 0; G.total = 0;
        for (int r = 0; r < W; r++) {
            uint64_t h[2]; memcpy(h, recv + (size_t)r * block, 16);
            G.counts[r] = h[0]; mx = std::max(mx, h[0]); G.total += h[0];
            if (h[1]) on_failed(r);
            G.block[r] = (const Rec*)(recv + (size_t)r * block + 16);
        }
        if (mx <= cap) return G;                                                    // (valid until the communicator's next gather)
        if (round) throw Error("gather_records: the second round did not fit (ranks disagree on the sizes)");
        cap = mx;
    }
}

void triangle_distributed(skh_ctx* ctx, Transport& T, const skh_sketch_set* L, double identity, int rescue_small, const skh_map_params& mp, uint32_t flags,
                          std::vector<uint32_t>& out_i, std::vector<uint32_t>& out_j, std::vector<skh_ani_result>& out_res, uint64_t* n_chained, skh_dist_stats* stats) {
    const int W = T.world, me = T.rank;
    if (W < 1 || me < 0 || me >= W || W > 255) throw std::invalid_argument("bad communicator");
    if (L->ctx != ctx) throw std::invalid_argument("the local sketch set belongs to another context");
    out_i.clear(); out_j.clear(); out_res.clear();
    skh_dist_stats st{};
    StageTrace tr(ctx);
    // the blocking exchange steps are timed on the host clock (the collectives may run on the transport's own stream); the sketch exchange is asynchronous
    // and adds only what the chaining had to wait for (exchange_end)
    std::chrono::steady_clock::time_point ex_t0; double exch_ms = 0;
    auto ex_begin = [&] { dsync(ctx->stream); ex_t0 = std::chrono::steady_clock::now(); };
    auto ex_end = [&] { dsync(ctx->stream); exch_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ex_t0).count(); };
    // Failures that only ONE rank can see (out of memory, a table overflow, a failed launch) must not leave the others waiting in the next
    // collective: every local phase runs under `local`, which keeps the first error; the status word travels with the next gather (or `agree`, an
    // 8-byte all-gather, where no gather follows) and makes every rank throw together, naming the rank that failed.  (Errors every rank derives
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
    struct ExchangeGuard {                                                          // the asynchronous sketch exchange is never left open, whatever unwinds
        Transport& T; skh_ctx* ctx; bool open = false;
        void close() { if (open) { open = false; T.exchange_end(ctx); } }
        ~ExchangeGuard() { try { close(); } catch (...) {} }
    };
    // What an open exchange reads and writes -- the count / offset arrays the transport keeps pointers to, the two device buffers -- is declared BEFORE the guard: whatever
    // unwinds, the guard closes the exchange (waiting for the worker thread / the second stream) while these still exist.
    std::vector<uint64_t> s_cnt, s_off, r_cnt, r_off;
    DBuf<uint32_t> send_buf, recv_buf;
    ExchangeGuard xg{T, ctx};
    auto stop_together = [&](const char* phase, int r) {                           // rank r reported a failure: every rank throws
        try { xg.close(); } catch (...) {}
        device_sync_all();                                                          // nothing queued may outlive the buffers the unwinding frees
        if (r == me) throw Error(std::string("distributed triangle, ") + phase + ": " + local_err);
        throw PeerError(std::string("distributed triangle, ") + phase + ": rank " + std::to_string(r) + " failed (its own error message says why); all ranks stop");
    };
    auto agree = [&](const char* phase) {
        uint64_t mine_ok = local_err.empty() ? 0 : 1; std::vector<uint64_t> all(W);
        T.all_gather(ctx, &mine_ok, all.data(), 8, false);
        for (int r = 0; r < W; r++) if (all[r]) stop_together(phase, r);
    };
    // ---- 1. who holds what, in ONE all-gather: per rank a head (genomes, seed positions, markers, contigs, c, k, marker_c, seeding mode, status), its per-genome
    // table and its contig lengths, in a block laid out by the capacities of the communicator's previous call (as gather_records does).  Round 5: the table also says,
    // per genome, how many of its markers fall into each of the W parts of the screen's key range (screen.hip part bounds) -- what the marker exchange of step 2 and
    // the marker-only sets on every rank are sized by.
    const uint32_t nL = L->n_genomes;
    const bool by_parts = W > 1 || ctx->tune.dist_key_range_w1;                       // (a world of one has nothing to cut: the row form, which is skh_triangle's own screen)
    const uint32_t PW = by_parts ? ((uint32_t)W + 1) / 2 : 0;                        // 64-bit words per genome that hold its W part counts (32 bits each)
    const uint32_t GF = 5 + PW; constexpr uint32_t HDR = 10;                        // per genome: seed positions, markers, contigs, total length, rank[, part counts]
    const uint64_t mine[8] = {nL, L->pos_off[nL], L->mk_off[nL], L->ctg_off[nL], L->params.c, L->params.k, L->params.marker_c, (uint64_t)L->params.seeding_mode | (L->wide ? 256u : 0u)};
    std::vector<uint64_t> part_lo; std::vector<uint32_t> part_n;                     // [g * W + r]: the stretch of local genome g's (sorted) marker set that lies in part r
    local([&] { if (by_parts && nL) screen_marker_parts(ctx, L, (uint32_t)W, part_lo, part_n); });   // (local phase 1; its status rides on the table gather)
    ex_begin();
    std::vector<uint64_t> cnt((size_t)W * 8), tab_all; size_t tab_words = 0;        // tab_all: the gathered blocks, tab_words each
    uint64_t max_n = 0, max_m = 0, max_c = 0;
    for (int round = 0;; round++) {
        const uint64_t cn = T.cap_n, cc = T.cap_c;
        tab_words = HDR + cn * GF + (cc + 1) / 2;
        std::vector<uint64_t> send(tab_words, 0); tab_all.assign((size_t)W * tab_words, 0);
        memcpy(send.data(), mine, sizeof(mine));
        send[8] = local_err.empty() ? 0 : 1;
        send[9] = ctx->tune.screen_cells ^ ((uint64_t)ctx->tune.dist_key_range_w1 << 63);   // what decides between the key-range and the row form of the screen (SKH_TUNE_*): must agree
        if (nL <= cn && mine[3] <= cc) {
            for (uint32_t g = 0; g < nL; g++) {
                uint64_t* f = send.data() + HDR + (size_t)g * GF;
                f[0] = L->pos_off[g + 1] - L->pos_off[g]; f[1] = L->mk_off[g + 1] - L->mk_off[g]; f[2] = L->ctg_off[g + 1] - L->ctg_off[g]; f[3] = L->total_len[g]; f[4] = L->rank[g];
                if (PW && !part_n.empty()) memcpy(f + 5, part_n.data() + (size_t)g * W, (size_t)W * 4);
            }
            if (!L->ctg_len.empty()) memcpy(send.data() + HDR + cn * GF, L->ctg_len.data(), L->ctg_len.size() * 4);
        }
        T.all_gather(ctx, send.data(), tab_all.data(), tab_words * 8, false);
        max_n = max_m = max_c = 0;
        for (int r = 0; r < W; r++) {
            const uint64_t* h = tab_all.data() + (size_t)r * tab_words;
            memcpy(cnt.data() + (size_t)r * 8, h, 64);
            if (h[4] != mine[4] || h[5] != mine[5] || h[6] != mine[6] || (h[7] & 255u) != (mine[7] & 255u))
                throw std::invalid_argument("the ranks sketched with different c / k / marker_c / seeding mode");
            if (h[9] != (ctx->tune.screen_cells ^ ((uint64_t)ctx->tune.dist_key_range_w1 << 63)))
                throw std::invalid_argument("the ranks run with different screen budgets (SKH_TUNE_SCREEN_CELLS / SKH_TUNE_DIST_KEY_RANGE_W1): they would take different collective sequences");
            max_n = std::max(max_n, h[0]); max_m = std::max(max_m, h[2]); max_c = std::max(max_c, h[3]);
        }
        for (int r = 0; r < W; r++) if (tab_all[(size_t)r * tab_words + 8]) stop_together("marker sets", r);
        if (max_n <= cn && max_c <= cc) break;
        if (round) throw Error("distributed triangle: the second table round did not fit (ranks disagree on the sizes)");
        T.cap_n = max_n; T.cap_c = max_c;
    }
    std::vector<uint64_t> base(W + 1, 0);
    // A rank with a wide set (a genome beyond 31-bit padded coordinates, internal.h): position records are not portable then (a wide genome's are
    // indices beside 64-bit coordinates), so ALL ranks exchange (position in contig, contig << 1 | canonical) -- the C ABI's form, 8 bytes instead
    // of 4 -- and the received genomes are indexed through the import path, which decides per genome from the contig lengths.
    bool wide_any = false;
    for (int r = 0; r < W; r++) { if (cnt[r * 8 + 7] & 256u) wide_any = true; base[r + 1] = base[r] + cnt[r * 8]; }   // (every rank sees the same table)
    const uint64_t N64 = base[W];
    if (N64 >= (1ull << 21)) throw std::invalid_argument("more than 2M genomes in one distributed triangle");
    const uint32_t N = (uint32_t)N64;
    st.n_genomes_total = N;
    std::vector<int> rank_of(N); std::vector<uint64_t> g_npos(N), g_nmk(N), g_nctg(N), g_len(N), g_rank(N);
    std::vector<uint32_t> cl_all; std::vector<uint64_t> g_ctg0(N);                  // contig lengths of all genomes, global genome order; g_ctg0: a genome's first
    std::vector<uint64_t> g_mine(N, 0);                                              // markers of genome g that fall into THIS rank's part of the key range
    {
        uint32_t g = 0;
        for (int r = 0; r < W; r++) {
            const uint64_t* blk = tab_all.data() + (size_t)r * tab_words;
            const uint32_t* cl = (const uint32_t*)(blk + HDR + T.cap_n * GF);
            uint64_t c0 = 0;
            for (uint64_t x = 0; x < cnt[r * 8]; x++, g++) {
                const uint64_t* f = blk + HDR + x * GF;
                rank_of[g] = r; g_npos[g] = f[0]; g_nmk[g] = f[1]; g_nctg[g] = f[2]; g_len[g] = f[3]; g_rank[g] = f[4]; g_ctg0[g] = cl_all.size();
                if (PW) g_mine[g] = ((const uint32_t*)(f + 5))[me];
                if (c0 + f[2] > cnt[r * 8 + 3]) throw Error("distributed triangle: a rank's contig table is inconsistent");
                cl_all.insert(cl_all.end(), cl + c0, cl + c0 + f[2]); c0 += f[2];
            }
        }
    }
    // ---- 2. the marker sets.  S: every genome's marker COUNT (what the rule's thresholds and the chaining's role decision take), and -- the row form only -- the sets
    // themselves.  The key-range form (the usual one) never needs a marker outside a rank's own part of the key range: ONE all-to-all moves, from every rank to every
    // rank, the stretches of its genomes' sorted marker sets that lie in the receiver's part -- a W-th of what the all-gather of rounds 3-4 moved to every rank
    // (config 4: 400 MB to each of eight ranks) -- straight into Sp, the marker-only set this rank screens.
    const bool key_range = by_parts && screen_parts_fit(ctx, N);
    skh_sketch_set S; S.ctx = ctx; S.params = L->params; S.n_genomes = N;
    S.mk_off.assign(N + 1, 0); for (uint32_t g = 0; g < N; g++) S.mk_off[g + 1] = S.mk_off[g] + g_nmk[g];
    skh_sketch_set Sp; Sp.ctx = ctx; Sp.params = L->params; Sp.n_genomes = N;
    const uint64_t MT = S.mk_off[N];
    if (key_range) {
        Sp.mk_off.assign(N + 1, 0); for (uint32_t g = 0; g < N; g++) Sp.mk_off[g + 1] = Sp.mk_off[g] + g_mine[g];
        std::vector<uint64_t> m_sc(W, 0), m_so(W, 0), m_rc(W, 0), m_ro(W, 0); uint64_t sw = 0;
        std::vector<uint64_t> seg;                                                  // 32-bit words: (source, destination, length) of every (genome, part) stretch
        for (int r = 0; r < W; r++) {
            m_so[r] = sw * 8;
            for (uint32_t g = 0; g < nL && !part_n.empty(); g++) {
                const uint64_t n = part_n[(size_t)g * W + r];
                if (n) { seg.insert(seg.end(), {part_lo[(size_t)g * W + r] * 2, sw * 2, n * 2}); sw += n; }
            }
            m_sc[r] = sw * 8 - m_so[r];
            m_ro[r] = Sp.mk_off[base[r]] * 8; m_rc[r] = (Sp.mk_off[base[r + 1]] - Sp.mk_off[base[r]]) * 8;
        }
        uint64_t* d_send = nullptr;
        local([&] {                                                                 // (local phase 2; agreed on in front of the exchange)
            S.markers.alloc(1); S.d_mk_off.alloc(N + 1); h2d(S.d_mk_off.p, S.mk_off.data(), (N + 1) * 8, ctx->stream);
            Sp.markers.alloc(Sp.mk_off[N] ? Sp.mk_off[N] : 1); Sp.d_mk_off.alloc(N + 1); h2d(Sp.d_mk_off.p, Sp.mk_off.data(), (N + 1) * 8, ctx->stream);
            d_send = ctx->arena.get<uint64_t>(sw ? sw : 1);
            copy_segments(ctx, (const uint32_t*)L->markers.p, (uint32_t*)d_send, seg);
            dsync(ctx->stream);
        });
        agree("marker sets");
        T.all_to_all_v(ctx, d_send, m_sc.data(), m_so.data(), Sp.markers.p, m_rc.data(), m_ro.data(), true);
        for (int r = 0; r < W; r++) if (r != me) st.marker_bytes_received += m_rc[r];
    } else {
        // the row form (a collection whose N x N count matrix is beyond the screen's budget; a world of one): every rank needs every marker
        uint64_t pad = std::max<uint64_t>(max_m, 1); uint64_t *d_mk_send = nullptr, *d_mk_recv = nullptr;
        local([&] {                                                                 // (local phase 2)
            d_mk_send = ctx->arena.get<uint64_t>(pad); d_mk_recv = ctx->arena.get<uint64_t>(pad * W);
            if (mine[2]) d2d(d_mk_send, L->markers.p, mine[2] * 8, ctx->stream);
            if (pad > mine[2]) dzero(d_mk_send + mine[2], (pad - mine[2]) * 8, ctx->stream);
            S.markers.alloc(MT ? MT : 1);
            S.d_mk_off.alloc(N + 1); h2d(S.d_mk_off.p, S.mk_off.data(), (N + 1) * 8, ctx->stream);
            dsync(ctx->stream);
        });
        agree("marker sets");
        T.all_gather(ctx, d_mk_send, d_mk_recv, pad * 8, true);
        for (int r = 0; r < W; r++) if (r != me) st.marker_bytes_received += cnt[r * 8 + 2] * 8;
        for (int r = 0; r < W; r++) if (cnt[r * 8 + 2]) d2d(S.markers.p + S.mk_off[base[r]], d_mk_recv + (uint64_t)r * pad, cnt[r * 8 + 2] * 8, ctx->stream);
        dsync(ctx->stream);
    }
    T.cap_m = std::max(T.cap_m, max_m);
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
    std::vector<uint32_t> pi, pj;
    if (key_range) {
        // ---- 3. screen by KEY RANGE (round 4; screen.hip): this rank sorts and walks the incidences of a W-th of the markers' leading 16 bases and gets partial counts
        // for all cells; the non-zero cells are gathered -- with the status of the phases so far -- and every rank adds them up and applies the rule to all rows
        // itself: the same candidate list everywhere, no list to gather.  (Cut by rows, every rank sorted and walked ALL incidences: the screen did not shrink with W.)
        // The cells (a 64-bit word each: i << 43 | j << 22 | count) never visit the host: they go from the arena into the communicator's own device buffers, blocks of
        // [count, status, cells ...] sized by what its previous call saw (a larger count makes every rank enlarge its buffers, agree, and go round once more).
        st.screen_by_key_range = 1;
        uint64_t* d_mine = nullptr; uint64_t n_mine = 0;
        local([&] {                                                                 // (local phase 3)
            Stopwatch sw(ctx, &ctx->timings.screen_ms);
            screen_partial_cells_dev(ctx, &Sp, 0u, 1u, &d_mine, &n_mine, screen_part_bound((uint32_t)me, (uint32_t)W), screen_part_bound((uint32_t)me + 1, (uint32_t)W));   // (Sp holds this rank's part of the key range and nothing else)
        });
        tr.mark("dist: screen, my key range");
        ex_begin();
        std::vector<uint64_t> heads(2 * (size_t)W);
        auto heads_say = [&](uint64_t& mx) {                                          // every rank reads the same heads: a failed rank stops all, the largest count
            mx = 0;
            for (int r = 0; r < W; r++) { if (heads[2 * r + 1]) stop_together("marker sets / screen", r); mx = std::max(mx, heads[2 * r]); }
        };
        auto grow = [&](uint64_t want) {                                              // (not one of the numbered local phases: it happens in a communicator's first call and when a collection grew)
            if (local_err.empty()) {
                try { T.cells_send.alloc(want + 2); T.cells_recv.alloc((want + 2) * (size_t)W); }
                catch (const std::exception& e) { local_err = e.what(); }
            }
            // every rank ends with the same capacity: the new one, or -- when any rank could not get its buffers -- none (the communicator's next call then starts
            // like its first on every rank, whatever this one throws)
            uint64_t mine_ok = local_err.empty() ? 0 : 1; std::vector<uint64_t> all(W);
            T.all_gather(ctx, &mine_ok, all.data(), 8, false);
            int failed = -1;
            for (int r = 0; r < W && failed < 0; r++) if (all[r]) failed = r;
            if (failed >= 0) { T.cells_send.release(); T.cells_recv.release(); T.cap_cells = 0; stop_together("cell buffers", failed); }
            T.cap_cells = want;
        };
        uint64_t mx = 0;
        if (T.cap_cells == 0) {                                                     // no buffers yet: the counts travel alone
            const uint64_t h[2] = {n_mine, local_err.empty() ? 0ull : 1ull};
            T.all_gather(ctx, h, heads.data(), 16, false);
            heads_say(mx);
            grow(std::max<uint64_t>(mx, 1));
        }
        for (int round = 0;; round++) {
            const uint64_t bw = T.cap_cells + 2;
            const uint64_t h[2] = {n_mine, local_err.empty() ? 0ull : 1ull};
            h2d(T.cells_send.p, h, 16, ctx->stream);
            if (n_mine && n_mine <= T.cap_cells && local_err.empty()) d2d(T.cells_send.p + 2, d_mine, n_mine * 8, ctx->stream);
            T.all_gather(ctx, T.cells_send.p, T.cells_recv.p, bw * 8, true);
            uint64_t* d_heads = ctx->arena.get<uint64_t>(2 * (size_t)W);
            SKH_LAUNCH(block_heads_kernel, (2 * W + 255) / 256, 256, 0, ctx->stream, (const uint64_t*)T.cells_recv.p, bw, (uint32_t)W, d_heads);
            check_launch("block_heads");
            d2h(heads.data(), d_heads, heads.size() * 8, ctx->stream);
            heads_say(mx);
            if (mx <= T.cap_cells) break;
            if (round) throw Error("distributed triangle: the second round of the cell gather did not fit (ranks disagree on the sizes)");
            grow(mx);
        }
        ex_end();
        ctx->arena.reset();
        local([&] {                                                                 // (local phase 4; a failure is agreed on in front of the sketch exchange)
            Stopwatch sw(ctx, &ctx->timings.screen_ms);
            screen_from_cells_dev(ctx, &S, T.cells_recv.p, (uint32_t)W, T.cap_cells + 2, mx, identity, rescue_small, pi, pj);
        });
        ctx->arena.reset();
        tr.mark("dist: cells gathered, candidates");
    } else {
        // ---- 3'. a collection whose dense count matrix is beyond the screen's budget: cut by rows (every rank walks all incidences and keeps its rows' cells), candidate lists gathered
        std::vector<uint32_t> my_i, my_j;
        local([&] {                                                                 // (local phase 3)
            if (rb[me + 1] <= rb[me]) return;
            Stopwatch sw(ctx, &ctx->timings.screen_ms);
            screen_pairs(ctx, &S, nullptr, identity, SKH_SCREEN_REFS, rescue_small, my_i, my_j, rb[me], rb[me + 1]);
        });
        local_phase++;                                                              // (the key-range form has two screen phases: the numbering of the later ones stays the same)
        ctx->arena.reset();
        tr.mark("dist: screen rows");
        struct Cand { uint32_t i, j; };
        ex_begin();
        {
            std::vector<Cand> mine_c(my_i.size());
            for (size_t x = 0; x < my_i.size(); x++) mine_c[x] = Cand{my_i[x], my_j[x]};
            const Gathered<Cand> G = gather_records(ctx, T, mine_c, !local_err.empty(), T.cap_pairs, [&](int r) { stop_together("marker sets / screen", r); });
            pi.clear(); pj.clear(); pi.reserve(G.total); pj.reserve(G.total);
            for (int r = 0; r < W; r++) for (uint64_t x = 0; x < G.counts[r]; x++) { pi.push_back(G.block[r][x].i); pj.push_back(G.block[r][x].j); }
        }
        ex_end();
    }
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
    std::vector<uint32_t> home_ids, away_ids;                                       // the genomes THIS rank chains: its own that stay / the ones it receives, ascending
    {
        std::vector<uint8_t> mark((size_t)N * W, 0);
        for (size_t p = 0; p < NP; p++) { mark[(size_t)pi[p] * W + owner[p]] = 1; mark[(size_t)pj[p] * W + owner[p]] = 1; }
        for (uint32_t g = 0; g < N; g++) {
            if (mark[(size_t)g * W + me]) (rank_of[g] == me ? home_ids : away_ids).push_back(g);
            for (int r = 0; r < W; r++)
                if (mark[(size_t)g * W + r] && rank_of[g] != r) {
                    if (rank_of[g] == me) send_to[r].push_back(g);
                    if (r == me) recv_from[rank_of[g]].push_back(g);
                }
        }
    }
    const uint32_t nR = (uint32_t)away_ids.size();
    st.n_genomes_received = nR;
    // ---- 7. the sketches that move travel ASYNCHRONOUSLY (RCCL: on the context's second stream; host collectives: on a thread), and meanwhile this rank
    // indexes the genomes of its own that it chains and chains its HOME pairs (both sketches its own).  The sets:
    //   H   the home side: the local set itself when its tables exist already, when nothing arrives, or with a wide set somewhere; otherwise a compact copy
    //       of the local genomes that stay (with deferred tables -- skh_sketch_genomes_ex -- a local sketch whose clusters went elsewhere is never indexed);
    //   Wr  the received genomes, in ascending global index, indexed when they have arrived; the AWAY pairs (a received sketch on either side) are
    //       chained over the two sets.
    std::unique_ptr<skh_sketch_set> Wh, Wr;
    std::vector<uint32_t> wk_set(N, 0), wk_index(N, 0xFFFFFFFFu);                   // global genome -> (0 = H, 1 = Wr, index in that set)
    const uint64_t NF = wide_any ? 3 : 2;                                           // 32-bit fields per seed position on the wire
    s_cnt.assign(W, 0); s_off.assign(W, 0); r_cnt.assign(W, 0); r_off.assign(W, 0);
    std::vector<uint64_t> seg_s, seg_g, seg_c, r_words(W, 0), recv_at(N, 0);
    uint64_t sw = 0, rw = 0;
    // send buffer per destination: [seeds of all its genomes][padded positions of all its genomes]  (32-bit words; with a wide set somewhere:
    // [seeds][positions in contig][contig << 1 | canonical])
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
    for (int r = 0; r < W; r++) {                                                   // recv_at: word offset of a received genome's seeds inside its source's block
        for (uint32_t g : recv_from[r]) { recv_at[g] = r_words[r]; r_words[r] += g_npos[g]; }
        r_off[r] = rw * 4; r_cnt[r] = r_words[r] * NF * 4; rw += r_words[r] * NF;
    }
    st.bytes_sent = sw * 4; st.bytes_received = rw * 4;
    uint32_t *d_send = nullptr, *d_recv = nullptr;
    // (the exchange buffers -- send_buf / recv_buf, declared in front of the guard -- live outside the arena: the chaining of the home pairs resets it while they are in flight)
    local([&] {                                                                     // (local phase 5)
        send_buf.alloc(sw + 1); recv_buf.alloc(rw + 1); d_send = send_buf.p; d_recv = recv_buf.p;
        copy_segments(ctx, L->p_seed.p, d_send, seg_s);
        if (wide_any) {
            const uint64_t PL = L->pos_off[nL];
            uint32_t* l_pos = ctx->arena.get<uint32_t>(PL + 1); uint32_t* l_cc = ctx->arena.get<uint32_t>(PL + 1);   // the local set's positions in the C ABI's form
            unpack_positions(ctx, L, 0, PL, l_pos, l_cc);
            copy_segments(ctx, l_pos, d_send, seg_g); copy_segments(ctx, l_cc, d_send, seg_c);
        } else copy_segments(ctx, L->p_g.p, d_send, seg_g);
        dsync(ctx->stream);
    });
    ctx->arena.reset();
    ex_begin();
    agree("sketch exchange buffers");
    ex_end();
    T.exchange_begin(ctx, d_send, s_cnt.data(), s_off.data(), d_recv, r_cnt.data(), r_off.data());
    xg.open = true;
    // -- the home side, while the sketches travel
    const bool compact_home = nR > 0 && !L->tables_built && !wide_any && !home_ids.empty();
    const skh_sketch_set* H = L;
    auto describe = [&](skh_sketch_set* X, const std::vector<uint32_t>& ids, uint32_t set_no) {   // host metadata of a set made of the listed global genomes
        const uint32_t n = (uint32_t)ids.size();
        X->ctx = ctx; X->params = L->params; X->n_genomes = n;
        X->rank.resize(n); X->pos_off.assign(n + 1, 0); X->mk_off.assign(n + 1, 0); X->ctg_off.assign(n + 1, 0); X->total_len.resize(n);
        for (uint32_t x = 0; x < n; x++) {
            const uint32_t g = ids[x];
            wk_set[g] = set_no; wk_index[g] = x;
            X->rank[x] = (uint32_t)g_rank[g]; X->total_len[x] = g_len[g];
            X->pos_off[x + 1] = X->pos_off[x] + g_npos[g]; X->mk_off[x + 1] = X->mk_off[x] + g_nmk[g]; X->ctg_off[x + 1] = X->ctg_off[x] + g_nctg[g];
            for (uint64_t c = 0; c < g_nctg[g]; c++) X->ctg_len.push_back(cl_all[g_ctg0[g] + c]);
        }
        finalize_metadata(X);
        X->markers.alloc(1);                                                        // (the chaining takes the marker COUNTS, chain.rs:625-649; the sets themselves stay in S)
    };
    local([&] {                                                                     // (local phase 6)
        if (!compact_home) { for (uint32_t g : home_ids) { wk_set[g] = 0; wk_index[g] = (uint32_t)(g - base[me]); } return; }
        Wh.reset(new skh_sketch_set());
        describe(Wh.get(), home_ids, 0);
        const uint64_t PW = Wh->pos_off[Wh->n_genomes];
        Wh->p_seed.alloc(PW ? PW : 1); Wh->p_g.alloc(PW);
        std::vector<uint64_t> seg;
        for (uint32_t x = 0; x < Wh->n_genomes; x++) { const uint32_t g = home_ids[x]; seg.insert(seg.end(), {L->pos_off[g - base[me]], Wh->pos_off[x], g_npos[g]}); }
        copy_segments(ctx, L->p_seed.p, Wh->p_seed.p, seg); copy_segments(ctx, L->p_g.p, Wh->p_g.p, seg);
        H = Wh.get();
    });
    local([&] { if (home_ids.empty()) return; Stopwatch swb(ctx, &ctx->timings.sketch_build_ms); ensure_tables(ctx, H); });   // (local phase 7)
    ctx->arena.reset();
    tr.mark("dist: home set indexed");
    // this rank's pairs: ref = genome i, query = genome j (triangle.rs:89-98); home pairs first
    std::vector<uint32_t> c_i, c_j; std::vector<uint8_t> c_away;
    for (size_t p = 0; p < NP; p++) {
        if (owner[p] != me) continue;
        c_i.push_back(pi[p]); c_j.push_back(pj[p]); c_away.push_back(rank_of[pi[p]] != me || rank_of[pj[p]] != me);
    }
    st.n_pairs_mine = c_i.size();
    std::vector<skh_ani_result> res(c_i.size());
    std::vector<uint32_t> sel_home, sel_away;
    for (uint32_t x = 0; x < c_i.size(); x++) (c_away[x] ? sel_away : sel_home).push_back(x);
    st.n_pairs_home = sel_home.size();
    // ties of switch_qr go by genome_rank on every rank: the sets made here carry no file names, and which rank chains a pair must not decide its orientation
    auto chain_selected = [&](const std::vector<uint32_t>& sel) {
        if (sel.empty()) return;
        const skh_sketch_set* sets[2] = {H, Wr ? Wr.get() : H};
        std::vector<uint32_t> rs(sel.size()), qs(sel.size()), rr(sel.size()), qq(sel.size());
        for (size_t x = 0; x < sel.size(); x++) {
            const uint32_t gi = c_i[sel[x]], gj = c_j[sel[x]];
            rs[x] = wk_set[gi]; rr[x] = wk_index[gi]; qs[x] = wk_set[gj]; qq[x] = wk_index[gj];
        }
        std::vector<skh_ani_result> part(sel.size());
        Stopwatch swc(ctx, &ctx->timings.chain_ms);
        chain_pairs(ctx, sets, Wr ? 2u : 1u, rs.data(), sets, Wr ? 2u : 1u, qs.data(), rr.data(), qq.data(), sel.size(), mp, part.data(), nullptr, true);
        for (size_t x = 0; x < sel.size(); x++) res[sel[x]] = part[x];
    };
    local([&] { chain_selected(sel_home); });                                       // (local phase 8)
    ctx->arena.reset();
    tr.mark("dist: home pairs chained");
    // -- the sketches have arrived (whatever happened above, the exchange is waited for: nobody may be left inside it)
    {
        const auto w0 = std::chrono::steady_clock::now();
        try { xg.close(); } catch (const std::exception& e) { if (local_err.empty()) local_err = e.what(); }
        dsync(ctx->stream);
        exch_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    }
    local([&] {                                                                     // (local phase 9)
        if (!nR) return;
        Wr.reset(new skh_sketch_set());
        describe(Wr.get(), away_ids, 1);
        const uint32_t nW = Wr->n_genomes; const uint64_t PW = Wr->pos_off[nW];
        Wr->p_seed.alloc(PW ? PW : 1);
        uint32_t *w_pos = nullptr, *w_cc = nullptr;                                 // wide_any: the positions in the C ABI's form, for the import path
        if (wide_any) { w_pos = ctx->arena.get<uint32_t>(PW + 1); w_cc = ctx->arena.get<uint32_t>(PW + 1); } else Wr->p_g.alloc(PW);
        std::vector<uint64_t> rs, rg, rc;
        for (uint32_t x = 0; x < nW; x++) {
            const uint32_t g = away_ids[x]; const uint64_t n = g_npos[g], dst = Wr->pos_off[x];
            const uint64_t b0 = r_off[rank_of[g]] / 4, rwd = r_words[rank_of[g]];
            rs.insert(rs.end(), {b0 + recv_at[g], dst, n}); rg.insert(rg.end(), {b0 + rwd + recv_at[g], dst, n});
            if (wide_any) rc.insert(rc.end(), {b0 + 2 * rwd + recv_at[g], dst, n});
        }
        copy_segments(ctx, d_recv, Wr->p_seed.p, rs);
        if (wide_any) { copy_segments(ctx, d_recv, w_pos, rg); copy_segments(ctx, d_recv, w_cc, rc); } else copy_segments(ctx, d_recv, Wr->p_g.p, rg);
        Stopwatch swb(ctx, &ctx->timings.sketch_build_ms);
        if (wide_any) build_sketch_tables(ctx, Wr.get(), w_pos, w_cc);             // (the position arrays live in the arena: the tables are made here)
        else ensure_tables(ctx, Wr.get());
    });
    ctx->arena.reset();
    tr.mark("dist: received sketches indexed");
    local([&] { chain_selected(sel_away); });                                       // (local phase 10)
    ctx->arena.reset();
    tr.mark("dist: away pairs chained");
    T.exchange_times(&st.exchange_async_us, &st.exchange_wait_us);
    // ---- 8. results (ani > 0.1, triangle.rs:99) gathered on every rank, sorted by (i, j)
    struct Row { uint32_t i, j; skh_ani_result r; };
    std::vector<Row> rows;
    for (size_t p = 0; p < res.size(); p++) if (local_err.empty() && res[p].ani > 0.1f) rows.push_back(Row{c_i[p], c_j[p], res[p]});
    ex_begin();
    if (flags & SKH_DIST_ROWS_TO_ROOT) {
        // "results gathered to rank 0".  The counts and the status of the chaining phases go round in a 16-byte gather (every rank must learn of a failure);
        // the rows themselves travel to rank 0 only -- an all-to-all in which every rank sends one block there -- and the other ranks return their own rows.
        uint64_t head[2] = {rows.size(), local_err.empty() ? 0ull : 1ull}; std::vector<uint64_t> heads(2 * (size_t)W);
        if (W == 1) { heads[0] = head[0]; heads[1] = head[1]; } else T.all_gather(ctx, head, heads.data(), 16, false);
        for (int r = 0; r < W; r++) if (heads[2 * (size_t)r + 1]) stop_together("seed tables / chaining", r);
        std::vector<uint64_t> sc(W, 0), so(W, 0), rc(W, 0), ro(W, 0); uint64_t total = 0;
        sc[0] = rows.size() * sizeof(Row);
        for (int r = 1; r < W; r++) so[r] = sc[0];                                   // (offsets in rank order without gaps: what MPI_Alltoallv-like callbacks expect)
        if (me == 0) for (int r = 0; r < W; r++) { rc[r] = heads[2 * (size_t)r] * sizeof(Row); ro[r] = total * sizeof(Row); total += heads[2 * (size_t)r]; }
        if (T.g_recv.size() < total * sizeof(Row) + 8) T.g_recv.resize(total * sizeof(Row) + 8);
        Row none{}; const void* sp = rows.empty() ? (const void*)&none : (const void*)rows.data();
        if (W > 1) T.all_to_all_v(ctx, sp, sc.data(), so.data(), T.g_recv.data(), rc.data(), ro.data(), false);
        ex_end();
        out_i.clear(); out_j.clear(); out_res.clear();
        if (me != 0 || W == 1) {                                                    // own rows, already in (i, j) order (the order of the candidate list)
            out_i.reserve(rows.size()); out_j.reserve(rows.size()); out_res.reserve(rows.size());
            for (const Row& row : rows) { out_i.push_back(row.i); out_j.push_back(row.j); out_res.push_back(row.r); }
        } else {
            out_i.reserve(total); out_j.reserve(total); out_res.reserve(total);
            const Row* all = (const Row*)T.g_recv.data();
            std::vector<uint64_t> cur(W, 0), first(W, 0);
            for (int r = 0; r < W; r++) first[r] = ro[r] / sizeof(Row);
            for (size_t p2 = 0; p2 < NP; p2++) {
                const int r = owner[p2]; const uint64_t x = cur[r];
                if (x >= heads[2 * (size_t)r]) continue;
                const Row& row = all[first[r] + x];
                if (row.i == pi[p2] && row.j == pj[p2]) { out_i.push_back(row.i); out_j.push_back(row.j); out_res.push_back(row.r); cur[r]++; }
            }
            if (out_i.size() != total) throw Error("distributed triangle: the gathered result rows do not follow the candidate list");
        }
    } else {
    const Gathered<Row> G = gather_records(ctx, T, rows, !local_err.empty(), T.cap_rows, [&](int r) { stop_together("seed tables / chaining", r); });
    ex_end();
    // The rows into (i, j) order.  Every rank sent its rows in the order of the candidate list, which every rank holds: one walk over that list with a cursor
    // per rank puts them in place (a sort of 95,000 rows of 72 bytes took milliseconds on every rank of config 4).
    out_i.clear(); out_j.clear(); out_res.clear(); out_i.reserve(G.total); out_j.reserve(G.total); out_res.reserve(G.total);
    {
        std::vector<uint64_t> cur(W, 0);
        for (size_t p2 = 0; p2 < NP; p2++) {
            const int r = owner[p2]; const uint64_t x = cur[r];
            if (x >= G.counts[r]) continue;
            const Row& row = G.block[r][x];
            if (row.i == pi[p2] && row.j == pj[p2]) { out_i.push_back(row.i); out_j.push_back(row.j); out_res.push_back(row.r); cur[r]++; }
        }
        if (out_i.size() != G.total) throw Error("distributed triangle: the gathered result rows do not follow the candidate list");
    }
    }
    ctx->timings.exchange_ms += (float)exch_ms;
    if (stats) *stats = st;
    tr.mark("dist: results gathered");
}

// One small all-gather of host memory, one of device memory and one sketch-exchange-shaped all-to-all through the communicator, every byte checked: run
// once after a communicator is made, where every rank can still choose another transport together (bench.py).
void comm_selftest(skh_ctx* ctx, Transport& T) {
    const int W = T.world, me = T.rank;
    const char* bad = nullptr;                                                      // (every collective is made before anything is thrown: no rank is left inside one)
    {   // host all-gather
        uint64_t mine[4]; std::vector<uint64_t> all((size_t)W * 4);
        for (int x = 0; x < 4; x++) mine[x] = (uint64_t)me * 1000 + x;
        T.all_gather(ctx, mine, all.data(), sizeof(mine), false);
        for (int r = 0; r < W; r++) for (int x = 0; x < 4; x++) if (all[(size_t)r * 4 + x] != (uint64_t)r * 1000 + x) bad = "communicator self-test: host all-gather returned wrong data";
    }
    {   // device all-gather
        const size_t n = 1024; std::vector<uint32_t> h(n), back(n * W);
        for (size_t x = 0; x < n; x++) h[x] = (uint32_t)(me * 100003u + x);
        DBuf<uint32_t> ds(n), dr(n * W);
        h2d(ds.p, h.data(), n * 4, ctx->stream); dsync(ctx->stream);
        T.all_gather(ctx, ds.p, dr.p, n * 4, true);
        d2h(back.data(), dr.p, n * W * 4, ctx->stream);
        for (int r = 0; r < W; r++) for (size_t x = 0; x < n; x++) if (back[(size_t)r * n + x] != (uint32_t)(r * 100003u + x)) bad = "communicator self-test: device all-gather returned wrong data";
    }
    for (int async = 0; async < 2; async++) {   // all-to-all with uneven shares (rank r sends 256 (1 + (r + q) % 3) words to rank q; nothing to itself when W > 1), blocking and asynchronous
        std::vector<uint64_t> sc(W), so(W), rc(W), ro(W); uint64_t sw = 0, rw = 0;
        auto words = [&](int r, int q) { return (uint64_t)((r == q && W > 1) ? 0 : 256 * (1 + (r + q) % 3)); };
        for (int q = 0; q < W; q++) { so[q] = sw * 4; sc[q] = words(me, q) * 4; sw += words(me, q); ro[q] = rw * 4; rc[q] = words(q, me) * 4; rw += words(q, me); }
        std::vector<uint32_t> h(sw + 1), back(rw + 1);
        for (int q = 0; q < W; q++) for (uint64_t x = 0; x < words(me, q); x++) h[so[q] / 4 + x] = (uint32_t)((me << 24) ^ (q << 16) ^ x);
        DBuf<uint32_t> ds(sw + 1), dr(rw + 1);
        h2d(ds.p, h.data(), sw * 4, ctx->stream); dsync(ctx->stream);
        if (async) { T.exchange_begin(ctx, ds.p, sc.data(), so.data(), dr.p, rc.data(), ro.data()); T.exchange_end(ctx); }
        else T.all_to_all_v(ctx, ds.p, sc.data(), so.data(), dr.p, rc.data(), ro.data(), true);
        d2h(back.data(), dr.p, rw * 4, ctx->stream);
        for (int q = 0; q < W; q++) for (uint64_t x = 0; x < words(q, me); x++)
            if (back[ro[q] / 4 + x] != (uint32_t)((q << 24) ^ (me << 16) ^ x)) bad = async ? "communicator self-test: asynchronous all-to-all returned wrong data" : "communicator self-test: all-to-all returned wrong data";
    }
    if (bad) throw Error(bad);
}

Transport* make_host_transport(const skh_host_collectives* hc, int rank, int world) {
    if (!hc || !hc->all_gather || !hc->all_to_all_v) throw std::invalid_argument("null collective");
    HostTransport* t = new HostTransport(); t->hc = *hc; t->rank = rank; t->world = world;
    return t;
}

}  // namespace skh
