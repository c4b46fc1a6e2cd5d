// capi.hip -- the extern "C" boundary declared in include/skani_hip.h.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>

#include "internal.h"

using namespace skh;

namespace {

// Every entry point runs its body through here: the calling thread is bound to the context's device and its small uploads to the context's pinned
// ring for the duration of the call; nothing unwinds across the boundary.
template <class F> int guarded(skh_ctx* ctx, F&& f) {
    try {
        if (ctx) { PinScope scope(ctx->device, &ctx->ring); if (!ctx->pending_sorts.empty()) reap_pending_sorts(ctx); f(); } else f();
        return SKH_OK;
    }
    catch (const std::bad_alloc&) { if (ctx) ctx->err = "out of host memory"; return SKH_ERR_NOMEM; }
    catch (const PeerError& e) { if (ctx) ctx->err = e.what(); return SKH_ERR_PEER; }
    catch (const Error& e) { if (ctx) ctx->err = e.what(); return SKH_ERR_DEVICE; }
    catch (const std::invalid_argument& e) { if (ctx) ctx->err = e.what(); return SKH_ERR_INVALID; }
    catch (const std::exception& e) { if (ctx) ctx->err = e.what(); return SKH_ERR_INTERNAL; }
    catch (...) { if (ctx) ctx->err = "unknown error"; return SKH_ERR_INTERNAL; }
}

void check_params(const skh_sketch_params* p) {
    if (!p) throw std::invalid_argument("null sketch params");
    if (p->k == 0 || p->k > 16) throw std::invalid_argument("k must be in 1..16 (seeding.rs:239)");
    if (p->c == 0) throw std::invalid_argument("c must be >= 1");
    if (p->c > p->marker_c) throw std::invalid_argument("c > marker_c is not allowed (params.rs:183-185)");
    if (p->seeding_mode > 1) throw std::invalid_argument("bad seeding_mode");
}

void load_model(skh_ctx* ctx, const char* path, GbdtModel& m) {
    FILE* f = fopen(path, "rb");
    if (!f) throw std::invalid_argument(std::string("cannot open model table ") + path);
    char magic[4]; uint32_t nt = 0, nf = 0, nn = 0; float sh = 0, bias = 0;
    bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "GBDT", 4) && fread(&nt, 4, 1, f) == 1 && fread(&nf, 4, 1, f) == 1 &&
              fread(&sh, 4, 1, f) == 1 && fread(&bias, 4, 1, f) == 1 && fread(&nn, 4, 1, f) == 1;
    std::vector<uint32_t> off(nt + 1); std::vector<GbdtModel::Node> nodes(nn);
    ok = ok && fread(off.data(), 4, nt + 1, f) == nt + 1 && fread(nodes.data(), sizeof(GbdtModel::Node), nn, f) == nn;
    fclose(f);
    if (!ok) throw std::invalid_argument(std::string("malformed model table ") + path);
    m.n_trees = nt; m.n_feat = nf; m.n_nodes = nn; m.shrinkage = sh; m.bias = bias;
    m.off.alloc(nt + 1); m.nodes.alloc(nn);
    h2d(m.off.p, off.data(), (nt + 1) * 4, ctx->stream); h2d(m.nodes.p, nodes.data(), nn * sizeof(GbdtModel::Node), ctx->stream);
    dsync(ctx->stream);
}

skh_sketch_set* new_sketch_set(skh_ctx* ctx, const skh_sketch_params& sp, uint32_t ng, const uint32_t* rank) {
    skh_sketch_set* ss = new skh_sketch_set();
    ss->ctx = ctx; ss->params = sp; ss->n_genomes = ng;
    ss->rank.resize(ng);
    for (uint32_t g = 0; g < ng; g++) ss->rank[g] = rank ? rank[g] : g;
    return ss;
}

}  // namespace

extern "C" {

int skh_ctx_create(int device, skh_ctx** out) {
    if (!out) return SKH_ERR_INVALID;
    *out = nullptr;
    skh_ctx* ctx = new (std::nothrow) skh_ctx();
    if (!ctx) return SKH_ERR_NOMEM;
    int rc = guarded(ctx, [&] {
        dev_open(device, &ctx->stream, &ctx->stream2);
        ctx->device = device; ctx->ring.s0 = ctx->stream; ctx->ring.s1 = ctx->stream2;
        ctx->scan_ticket.alloc(2); dzero(ctx->scan_ticket.p, 8, ctx->stream); dsync(ctx->stream);
        auto env = [](const char* n, uint64_t dflt) { const char* v = getenv(n); return v && *v ? (uint64_t)strtoull(v, nullptr, 10) : dflt; };
        ctx->tune.seed_scratch_bytes = env("SKH_TUNE_SEED_SCRATCH_BYTES", ctx->tune.seed_scratch_bytes); ctx->tune.seed_scratch_fixed = getenv("SKH_TUNE_SEED_SCRATCH_BYTES") != nullptr;
        ctx->tune.seed_tile_cap = (uint32_t)env("SKH_TUNE_SEED_TILE_CAP", ctx->tune.seed_tile_cap);
        ctx->tune.screen_cells = env("SKH_TUNE_SCREEN_CELLS", ctx->tune.screen_cells);
        ctx->tune.chain_anchors = env("SKH_TUNE_CHAIN_ANCHORS", ctx->tune.chain_anchors);
        ctx->tune.chain_super_tiles = (uint32_t)env("SKH_TUNE_CHAIN_SUPER_TILES", ctx->tune.chain_super_tiles);
        ctx->tune.chain_dp_lds_slots = (uint32_t)env("SKH_TUNE_CHAIN_DP_LDS_SLOTS", ctx->tune.chain_dp_lds_slots);
        ctx->tune.build_match_cap = (uint32_t)env("SKH_TUNE_BUILD_MATCH_CAP", ctx->tune.build_match_cap);
        ctx->tune.marker_lds_max = (uint32_t)env("SKH_TUNE_MARKER_LDS_MAX", ctx->tune.marker_lds_max);
        ctx->tune.build_slice_max = (uint32_t)env("SKH_TUNE_BUILD_SLICE_MAX", ctx->tune.build_slice_max);
        ctx->tune.join_bitmap_words = (uint32_t)env("SKH_TUNE_JOIN_BITMAP_WORDS", ctx->tune.join_bitmap_words);
        ctx->tune.screen_planes = (uint32_t)env("SKH_TUNE_SCREEN_PLANES", ctx->tune.screen_planes);
        ctx->tune.screen_count_rows = (uint32_t)env("SKH_TUNE_SCREEN_COUNT_ROWS", ctx->tune.screen_count_rows);
        ctx->tune.screen_col_order = (uint32_t)env("SKH_TUNE_SCREEN_COL_ORDER", ctx->tune.screen_col_order);
        ctx->tune.marker_gate = (uint32_t)env("SKH_TUNE_MARKER_GATE", ctx->tune.marker_gate);
        ctx->tune.screen_cells_dense = (uint32_t)env("SKH_TUNE_SCREEN_CELLS_DENSE", 0);
        ctx->tune.build_resalt_all = (uint32_t)env("SKH_TUNE_BUILD_RESALT_ALL", 0);
        ctx->tune.screen_sort_radix = (uint32_t)env("SKH_TUNE_SCREEN_SORT_RADIX", 0);
        ctx->tune.skeys_avg = (uint32_t)env("SKH_TUNE_SKEYS_AVG", ctx->tune.skeys_avg);
        ctx->tune.skeys_cap = (uint32_t)env("SKH_TUNE_SKEYS_CAP", 0);
        ctx->tune.wide_sweep_dp = (uint32_t)env("SKH_TUNE_WIDE_SWEEP_DP", 0);
        ctx->tune.scan_one_max = env("SKH_TUNE_SCAN_ONE_MAX", ctx->tune.scan_one_max); ctx->tune.scan_two_max = env("SKH_TUNE_SCAN_TWO_MAX", ctx->tune.scan_two_max);
        ctx->tune.dist_fail = (uint32_t)env("SKH_TUNE_DIST_FAIL", 0);
        ctx->tune.dist_key_range_w1 = (uint32_t)env("SKH_TUNE_DIST_KEY_RANGE_W1", 0);
        ctx->tune.greedy_big_min = (uint32_t)env("SKH_TUNE_GREEDY_BIG_MIN", ctx->tune.greedy_big_min);
        ctx->tune.wide_span = std::min<uint64_t>(ctx->tune.wide_span, env("SKH_TUNE_WIDE_SPAN", ctx->tune.wide_span));
        ctx->tune.greedy_len_limit = (uint32_t)std::min<uint64_t>(0x10000, env("SKH_TUNE_GREEDY_LEN_LIMIT", 0x10000));
    });
    if (rc != SKH_OK) { delete ctx; return rc; }
    *out = ctx;
    return SKH_OK;
}

void skh_ctx_destroy(skh_ctx* ctx) {
    if (!ctx) return;
    dev_drain(ctx->device, ctx->stream, ctx->stream2);
    ctx->arena.release_all();
    ctx->model_c125 = GbdtModel(); ctx->model_c200 = GbdtModel(); ctx->scan_ticket.release(); ctx->part_cnt.release();
    dcache_trim();                                   // hand the allocator's idle blocks back to the driver
    dev_close(ctx->stream, ctx->stream2);
    delete ctx;
}

const char* skh_last_error(const skh_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int skh_device_memory(uint64_t* live_bytes, uint64_t* idle_bytes, int trim) {
    try {
        if (trim) dcache_trim();
        size_t live = 0, idle = 0; dcache_stats(&live, &idle);
        if (live_bytes) *live_bytes = live;
        if (idle_bytes) *idle_bytes = idle;
        return SKH_OK;
    } catch (...) { return SKH_ERR_INTERNAL; }
}
void skh_free(void* p) { free(p); }

int skh_load_models(skh_ctx* ctx, const char* p125, const char* p200) {
    if (!ctx || !p125 || !p200) return SKH_ERR_INVALID;
    return guarded(ctx, [&] { load_model(ctx, p125, ctx->model_c125); load_model(ctx, p200, ctx->model_c200); });
}

int skh_genomes_pack(skh_ctx* ctx, const uint8_t* bases, const uint64_t* contig_off, const uint32_t* contig_genome, uint32_t n_contigs,
                     uint32_t n_genomes, int on_device, int seeding_mode, skh_genome_set** out) {
    if (!ctx || !out || (n_contigs && (!bases || !contig_off || !contig_genome))) return SKH_ERR_INVALID;
    *out = nullptr;
    skh_genome_set* gs = nullptr;
    int rc = guarded(ctx, [&] {
        if (seeding_mode != SKH_SEED_SCALAR && seeding_mode != SKH_SEED_AVX2) throw std::invalid_argument("bad seeding_mode");
        gs = new skh_genome_set();
        gs->ctx = ctx; gs->seeding_mode = seeding_mode;
        std::vector<uint64_t> len(n_contigs);
        for (uint32_t i = 0; i < n_contigs; i++) {
            if (contig_off[i + 1] < contig_off[i]) throw std::invalid_argument("contig_off must ascend");
            if (contig_genome[i] >= n_genomes || (i && contig_genome[i] < contig_genome[i - 1])) throw std::invalid_argument("contig_genome must be non-decreasing and < n_genomes");
            len[i] = contig_off[i + 1] - contig_off[i];
        }
        Stopwatch sw(ctx, &ctx->timings.pack_ms);
        genomes_begin(ctx, gs, n_contigs ? contig_off[n_contigs] - contig_off[0] : 0, n_contigs, n_genomes);
        genomes_append(ctx, gs, bases, contig_off, len.data(), contig_genome, n_contigs, on_device, nullptr);
        genomes_finish(ctx, gs);
    });
    ctx->arena.reset();
    if (rc != SKH_OK) { if (gs) device_sync_all(); delete gs; return rc; }
    *out = gs;
    return SKH_OK;
}

void* skh_host_alloc(uint64_t bytes) { try { return pin_alloc((size_t)bytes); } catch (...) { return nullptr; } }
void skh_host_free(void* p) { if (p) pin_free(p); }

int skh_genomes_begin(skh_ctx* ctx, uint64_t max_bases, uint32_t max_contigs, uint32_t n_genomes, int seeding_mode, skh_genome_set** out) {
    if (!ctx || !out) return SKH_ERR_INVALID;
    *out = nullptr;
    skh_genome_set* gs = nullptr;
    int rc = guarded(ctx, [&] {
        if (seeding_mode != SKH_SEED_SCALAR && seeding_mode != SKH_SEED_AVX2) throw std::invalid_argument("bad seeding_mode");
        gs = new skh_genome_set();
        gs->ctx = ctx; gs->seeding_mode = seeding_mode;
        genomes_begin(ctx, gs, max_bases, max_contigs, n_genomes);
    });
    if (rc != SKH_OK) { delete gs; return rc; }
    *out = gs;
    return SKH_OK;
}

int skh_genomes_append(skh_genome_set* gs, const uint8_t* bases, const uint64_t* contig_start, const uint64_t* contig_len, const uint32_t* contig_genome,
                       uint32_t n_contigs, int on_device, uint64_t* ticket) {
    if (!gs || (n_contigs && (!bases || !contig_start || !contig_len || !contig_genome))) return SKH_ERR_INVALID;
    skh_ctx* ctx = gs->ctx;
    int rc = guarded(ctx, [&] {
        DevEvent* ev = nullptr; size_t id = 0;
        { std::lock_guard<std::mutex> lk(gs->copied_mu); gs->copied.emplace_back(new DevEvent()); ev = gs->copied.back().get(); id = gs->copied.size() - 1; }
        genomes_append(ctx, gs, bases, contig_start, contig_len, contig_genome, n_contigs, on_device, ev);
        if (ticket) *ticket = id;
    });
    if (rc != SKH_OK) device_sync_all();
    return rc;                                                                   // (the arena's small tables of this batch are recycled by skh_genomes_finish)
}

int skh_genomes_wait(skh_genome_set* gs, uint64_t ticket) {                      // any thread, also while another thread appends: touches nothing but the batch's event
    if (!gs) return SKH_ERR_INVALID;
    std::shared_ptr<DevEvent> ev;
    { std::lock_guard<std::mutex> lk(gs->copied_mu); if (ticket < gs->copied.size()) ev = gs->copied[ticket]; }
    if (!ev) return SKH_ERR_INVALID;
    try { ev->wait(); return SKH_OK; } catch (...) { return SKH_ERR_DEVICE; }
}

int skh_genomes_finish(skh_genome_set* gs) {
    if (!gs) return SKH_ERR_INVALID;
    skh_ctx* ctx = gs->ctx;
    int rc = guarded(ctx, [&] { Stopwatch sw(ctx, &ctx->timings.pack_ms); genomes_finish(ctx, gs); std::lock_guard<std::mutex> lk(gs->copied_mu); gs->copied.clear(); });
    if (rc != SKH_OK) device_sync_all();
    ctx->arena.reset();
    return rc;
}

void skh_genomes_destroy(skh_genome_set* gs) { delete gs; }
uint64_t skh_genomes_total_bases(const skh_genome_set* gs) { return gs ? gs->total_bases : 0; }

int skh_sketch_genomes(skh_ctx* ctx, const skh_genome_set* gs_c, const skh_sketch_params* sp, const uint32_t* genome_rank, skh_sketch_set** out) {
    return skh_sketch_genomes_ex(ctx, gs_c, sp, genome_rank, 0, out);
}

int skh_sketch_build_tables(skh_ctx* ctx, skh_sketch_set* ss) {
    if (!ctx || !ss) return SKH_ERR_INVALID;
    const int rc = guarded(ctx, [&] { Stopwatch sw(ctx, &ctx->timings.sketch_build_ms); ensure_tables(ctx, ss); });
    ctx->arena.reset();
    return rc;
}

int skh_sketch_genomes_ex(skh_ctx* ctx, const skh_genome_set* gs_c, const skh_sketch_params* sp, const uint32_t* genome_rank, uint32_t flags, skh_sketch_set** out) {
    if (!ctx || !gs_c || !out) return SKH_ERR_INVALID;
    *out = nullptr;
    skh_genome_set* gs = const_cast<skh_genome_set*>(gs_c);
    skh_sketch_set* ss = nullptr;
    StageTrace tr(ctx);
    int rc = guarded(ctx, [&] {
        check_params(sp);
        if (gs->open) throw std::invalid_argument("the genome set is still being filled: call skh_genomes_finish first");
        if ((int)sp->seeding_mode != gs->seeding_mode) throw std::invalid_argument("genome set was packed for the other seeding_mode");
        ss = new_sketch_set(ctx, *sp, gs->n_genomes, genome_rank);
        ss->compact = (flags & SKH_SKETCH_COMPACT) != 0;
        const uint32_t ng = gs->n_genomes;
        // The set's host metadata (contig tables, padded contig starts, length quantiles: ~0.1 ms per 1000 genomes) is made WHILE the seeding kernel runs, not in front of
        // it -- the device used to idle for it at the start of every sketch call.  The one thing the seeding needs beforehand is whether the set is a wide one.
        bool wide = false;
        {
            std::vector<uint64_t> span(ng, CTG_PAD);
            for (uint32_t i = 0; i < gs->n_contigs; i++) span[gs->contigs[i].genome] += (uint64_t)gs->contigs[i].len + CTG_PAD;
            for (uint32_t g = 0; g < ng && !wide; g++) wide = span[g] >= ctx->tune.wide_span;
        }
        const std::function<void()> metadata = [&] {
            ss->ctg_off = gs->genome_contig_off; ss->ctg_len.resize(gs->n_contigs); ss->total_len.assign(ng, 0);
            for (uint32_t i = 0; i < gs->n_contigs; i++) { ss->ctg_len[i] = gs->contigs[i].len; ss->total_len[gs->contigs[i].genome] += gs->contigs[i].len; }
            finalize_metadata(ss);
            if (ss->wide != wide) throw Error("internal: the set's width was misjudged ahead of its metadata");
        };
        tr.mark("sketch: set made");
        SeedOutput so;
        // Phase times from three events on the main stream, read when the call is over: nothing waits between the seeding's last kernel (the
        // compaction, ~0.3 ms) and the table build, whose host-side tables are prepared while that kernel runs.
        DevEvent ev[3];
        ev[0].record(ctx->stream);
        auto book = [&] {                                                             // (the caller has synchronised)
            ev[2].record(ctx->stream); ev[2].wait();
            ctx->timings.seed_ms += DevEvent::ms(ev[0], ev[1]); ctx->timings.sketch_build_ms += DevEvent::ms(ev[1], ev[2]);
        };
        seed_genomes(ctx, gs, *sp, so, true, wide, &metadata);
        ev[1].record(ctx->stream);
        tr.mark("sketch: seeded");
        // (from here on kernels may still be queued that read the arena's scratch and write `so`: no buffer goes away on an error before they are done)
        struct TailGuard { bool armed = true; ~TailGuard() { if (armed) device_sync_all(); } } tail_guard;
        ss->p_seed = std::move(so.seed); ss->p_g = std::move(so.g); ss->p_g64 = std::move(so.g64); ss->pos_off = so.pos_off;
        // the seed tables are queued on the main stream; the marker sets (a sort and a few small kernels, with a read-back of their own) and the
        // screen's sorted incidence list are built on the second stream meanwhile
        if (flags & SKH_SKETCH_DEFER_TABLES) {                                       // markers only; the tables are built where (and if) the sketches are chained
            ss->dist_off.assign(ss->n_genomes + 1, 0);
            upload_set_offsets(ctx, ss);
            if (flags & SKH_SKETCH_NO_SCREEN_INDEX) build_markers(ctx, ss, so.markers_raw, so.mk_off);
            else { build_markers(ctx, ss, so.markers_raw, so.mk_off); prepare_screen_keys(ctx, ss); }
            dsync(ctx->stream);
            book(); tail_guard.armed = false;
            return;
        }
        // the marker sets' kernel first, on the second stream, with an event behind it that build_tables_kernel waits for (internal.h MarkerBuild)
        MarkerBuild mb;
        ev[1].make_wait(ctx->stream2);                                                // the raw markers come out of the compaction kernel
        if (ctx->tune.marker_gate) {
            std::swap(ctx->stream, ctx->stream2);
            try { build_markers_begin(ctx, ss, so.markers_raw, so.mk_off, mb); }
            catch (...) { std::swap(ctx->stream, ctx->stream2); device_sync_all(); throw; }
            std::swap(ctx->stream, ctx->stream2);
        }
        TableBuild tb = build_sketch_tables_begin(ctx, ss, nullptr, nullptr, mb.launched ? &mb.done : nullptr);
        tr.mark("sketch: tables queued");
        // (letting the second stream start only beside the table build's big kernels -- instead of beside the small copies and fills in front of them,
        //  which it holds back by ~150 us -- was measured in round 3: the marker-set kernel then starves beside build_tables_kernel, 1.03 instead of 0.31 ms,
        //  and the sketch phase grows from 1.60 to 1.81 ms)
        std::swap(ctx->stream, ctx->stream2);
        try {                                                                          // + the screen's sorted incidence list: counted beside the marker sets, its last kernel queued, not waited for
            ScreenKeysPlan plan; uint32_t plan_max = 0;
            build_markers(ctx, ss, so.markers_raw, so.mk_off, &plan, &plan_max, &mb);
            prepare_screen_keys(ctx, ss, /*async=*/true, &plan, plan_max);
        }
        catch (...) { std::swap(ctx->stream, ctx->stream2); device_sync_all(); throw; }
        std::swap(ctx->stream, ctx->stream2);
        tr.mark("sketch: markers + screen index");
        if (!ss->compact) prepare_halves(ctx, ss, /*ahead=*/true);                    // (host work + an upload while the build's kernels run; chain.hip)
        if (build_sketch_tables_finish(ctx, ss, tb) || ss->compact) prepare_halves(ctx, ss, false, /*again=*/true);
        book(); tail_guard.armed = false;
        tr.mark("sketch: tables finished");
    });
    ctx->arena.reset();
    tr.mark("sketch: arena reset");
    if (rc != SKH_OK) { delete ss; return rc; }
    *out = ss;
    return SKH_OK;
}

int skh_sketch_batch(skh_ctx* ctx, const uint8_t* bases, const uint64_t* contig_off, const uint32_t* contig_genome, uint32_t n_contigs,
                     uint32_t n_genomes, const skh_sketch_params* sp, const uint32_t* genome_rank, skh_sketch_set** out) {
    if (!ctx || !sp || !out) return SKH_ERR_INVALID;
    skh_genome_set* gs = nullptr;
    int rc = skh_genomes_pack(ctx, bases, contig_off, contig_genome, n_contigs, n_genomes, 0, (int)sp->seeding_mode, &gs);
    if (rc != SKH_OK) return rc;
    rc = skh_sketch_genomes(ctx, gs, sp, genome_rank, out);
    skh_genomes_destroy(gs);
    return rc;
}

void skh_sketch_set_destroy(skh_sketch_set* ss) { delete ss; }

int skh_sketch_set_names(skh_sketch_set* ss, const char* const* names) {
    if (!ss || !names) return SKH_ERR_INVALID;
    return guarded(ss->ctx, [&] { ss->names.resize(ss->n_genomes); for (uint32_t g = 0; g < ss->n_genomes; g++) ss->names[g] = names[g] ? names[g] : ""; });
}
uint32_t skh_sketch_n_genomes(const skh_sketch_set* ss) { return ss ? ss->n_genomes : 0; }
int skh_sketch_is_wide(const skh_sketch_set* ss) { return ss && ss->wide ? 1 : 0; }

int skh_sketch_sizes(const skh_sketch_set* ss, uint32_t g, uint64_t* n_pos, uint64_t* n_distinct, uint64_t* n_markers, uint32_t* n_contigs, uint64_t* total_len) {
    if (!ss || g >= ss->n_genomes) return SKH_ERR_INVALID;
    if (n_distinct && !ss->tables_built) { const int rc = skh_sketch_build_tables(ss->ctx, const_cast<skh_sketch_set*>(ss)); if (rc != SKH_OK) return rc; }   // the distinct-seed counts come out of the table build
    if (n_pos) *n_pos = ss->pos_off[g + 1] - ss->pos_off[g];
    if (n_distinct) *n_distinct = ss->dist_off[g + 1] - ss->dist_off[g];
    if (n_markers) *n_markers = ss->mk_off[g + 1] - ss->mk_off[g];
    if (n_contigs) *n_contigs = (uint32_t)(ss->ctg_off[g + 1] - ss->ctg_off[g]);
    if (total_len) *total_len = ss->total_len[g];
    return SKH_OK;
}

int skh_sketch_export(const skh_sketch_set* ss, uint32_t g, uint32_t* seed, uint32_t* pos, uint32_t* cc, uint64_t* markers, uint32_t* contig_lengths) {
    if (!ss || g >= ss->n_genomes) return SKH_ERR_INVALID;
    skh_ctx* ctx = ss->ctx;
    const int rc = guarded(ctx, [&] {
        const uint64_t p0 = ss->pos_off[g], np = ss->pos_off[g + 1] - p0, m0 = ss->mk_off[g], nm = ss->mk_off[g + 1] - m0;
        if (seed) d2h(seed, ss->p_seed.p + p0, np * 4, ctx->stream);
        if ((pos || cc) && np) {
            uint32_t* tp = ctx->arena.get<uint32_t>(np); uint32_t* tc = ctx->arena.get<uint32_t>(np);
            unpack_positions(ctx, ss, p0, np, tp, tc);
            if (pos) d2h(pos, tp, np * 4, ctx->stream);
            if (cc) d2h(cc, tc, np * 4, ctx->stream);
        }
        if (markers) d2h(markers, ss->markers.p + m0, nm * 8, ctx->stream);
        if (contig_lengths) memcpy(contig_lengths, ss->ctg_len.data() + ss->ctg_off[g], (ss->ctg_off[g + 1] - ss->ctg_off[g]) * 4);
        dsync(ctx->stream);
    });
    ctx->arena.reset();
    return rc;
}

int skh_sketch_import_flat(skh_ctx* ctx, const skh_sketch_params* sp, uint32_t ng, int on_device, const uint64_t* pos_off, const uint32_t* seed,
                           const uint32_t* pos, const uint32_t* cc, const uint64_t* marker_off, const uint64_t* markers, const uint64_t* contig_off,
                           const uint32_t* contig_lengths, const uint64_t* total_len, const uint32_t* genome_rank, skh_sketch_set** out) {
    if (!ctx || !out || !pos_off || !marker_off || !contig_off || !total_len) return SKH_ERR_INVALID;
    *out = nullptr;
    skh_sketch_set* ss = nullptr;
    int rc = guarded(ctx, [&] {
        check_params(sp);
        ss = new_sketch_set(ctx, *sp, ng, genome_rank);
        ss->pos_off.assign(pos_off, pos_off + ng + 1); ss->mk_off.assign(marker_off, marker_off + ng + 1);
        ss->ctg_off.assign(contig_off, contig_off + ng + 1);
        ss->ctg_len.assign(contig_lengths, contig_lengths + contig_off[ng]);
        ss->total_len.assign(total_len, total_len + ng);
        finalize_metadata(ss);
        const uint64_t P = pos_off[ng], M = marker_off[ng];
        if ((P && (!seed || !pos || !cc)) || (M && !markers)) throw std::invalid_argument("null sketch array");
        ss->p_seed.alloc(P); ss->markers.alloc(M);
        auto put = [&](void* d, const void* src, size_t n) { if (on_device) d2d(d, src, n, ctx->stream); else h2d(d, src, n, ctx->stream); };
        put(ss->p_seed.p, seed, P * 4); put(ss->markers.p, markers, M * 8);
        ss->d_mk_off.alloc(ng + 1); h2d(ss->d_mk_off.p, ss->mk_off.data(), (ng + 1) * 8, ctx->stream);
        const uint32_t *dpos = pos, *dcc = cc;
        if (!on_device && P) {
            uint32_t* tp = ctx->arena.get<uint32_t>(P); uint32_t* tc = ctx->arena.get<uint32_t>(P);
            h2d(tp, pos, P * 4, ctx->stream); h2d(tc, cc, P * 4, ctx->stream); dpos = tp; dcc = tc;
        }
        build_sketch_tables(ctx, ss, dpos, dcc);
    });
    ctx->arena.reset();
    if (rc != SKH_OK) { delete ss; return rc; }
    *out = ss;
    return SKH_OK;
}

int skh_sketch_import(skh_ctx* ctx, const skh_sketch_params* sp, uint32_t ng, const uint64_t* pos_off, const uint32_t* seed, const uint32_t* pos,
                      const uint32_t* cc, const uint64_t* marker_off, const uint64_t* markers, const uint64_t* contig_off,
                      const uint32_t* contig_lengths, const uint64_t* total_len, const uint32_t* genome_rank, skh_sketch_set** out) {
    return skh_sketch_import_flat(ctx, sp, ng, 0, pos_off, seed, pos, cc, marker_off, markers, contig_off, contig_lengths, total_len, genome_rank, out);
}

int skh_sketch_totals(const skh_sketch_set* ss, uint64_t* n_pos, uint64_t* n_markers, uint64_t* n_contigs) {
    if (!ss) return SKH_ERR_INVALID;
    if (n_pos) *n_pos = ss->pos_off[ss->n_genomes];
    if (n_markers) *n_markers = ss->mk_off[ss->n_genomes];
    if (n_contigs) *n_contigs = ss->ctg_off[ss->n_genomes];
    return SKH_OK;
}

int skh_sketch_export_flat(const skh_sketch_set* ss, int on_device, uint32_t* seed, uint32_t* pos, uint32_t* cc, uint64_t* markers, uint64_t* pos_off,
                           uint64_t* marker_off, uint64_t* contig_off, uint32_t* contig_lengths, uint64_t* total_len, uint32_t* genome_rank) {
    if (!ss) return SKH_ERR_INVALID;
    skh_ctx* ctx = ss->ctx;
    const int rc = guarded(ctx, [&] {
        const uint32_t ng = ss->n_genomes;
        const uint64_t P = ss->pos_off[ng], M = ss->mk_off[ng];
        auto get = [&](void* dst, const void* src, size_t n) { if (!dst || !n) return; if (on_device) d2d(dst, src, n, ctx->stream); else d2h(dst, src, n, ctx->stream); };
        get(seed, ss->p_seed.p, P * 4); get(markers, ss->markers.p, M * 8);
        if ((pos || cc) && P) {
            if (on_device) unpack_positions(ctx, ss, 0, P, pos, cc);
            else {
                uint32_t* tp = ctx->arena.get<uint32_t>(P); uint32_t* tc = ctx->arena.get<uint32_t>(P);
                unpack_positions(ctx, ss, 0, P, tp, tc);
                get(pos, tp, P * 4); get(cc, tc, P * 4);
            }
        }
        if (pos_off) memcpy(pos_off, ss->pos_off.data(), (ng + 1) * 8);
        if (marker_off) memcpy(marker_off, ss->mk_off.data(), (ng + 1) * 8);
        if (contig_off) memcpy(contig_off, ss->ctg_off.data(), (ng + 1) * 8);
        if (contig_lengths && !ss->ctg_len.empty()) memcpy(contig_lengths, ss->ctg_len.data(), ss->ctg_len.size() * 4);
        if (total_len && ng) memcpy(total_len, ss->total_len.data(), ng * 8);
        if (genome_rank && ng) memcpy(genome_rank, ss->rank.data(), ng * 4);
        dsync(ctx->stream);
    });
    ctx->arena.reset();
    return rc;
}

int skh_screen(skh_ctx* ctx, const skh_sketch_set* refs, const skh_sketch_set* queries, double identity, int rule, int rescue_small,
               uint32_t** pair_first, uint32_t** pair_second, uint64_t* n_pairs) {
    if (!ctx || !refs || !pair_first || !pair_second || !n_pairs) return SKH_ERR_INVALID;
    *pair_first = *pair_second = nullptr; *n_pairs = 0;
    int rc = guarded(ctx, [&] {
        if (rule < 0 || rule > 2) throw std::invalid_argument("bad screen rule");
        std::vector<uint32_t> a, b;
        { Stopwatch sw(ctx, &ctx->timings.screen_ms); screen_pairs(ctx, refs, queries, identity, rule, rescue_small, a, b); }
        uint32_t* pa = (uint32_t*)malloc((a.size() + 1) * 4); uint32_t* pb = (uint32_t*)malloc((b.size() + 1) * 4);
        if (!pa || !pb) { free(pa); free(pb); throw std::bad_alloc(); }
        memcpy(pa, a.data(), a.size() * 4); memcpy(pb, b.data(), b.size() * 4);
        *pair_first = pa; *pair_second = pb; *n_pairs = a.size();
    });
    ctx->arena.reset();
    return rc;
}

int skh_screen_rows(skh_ctx* ctx, const skh_sketch_set* set, uint32_t row0, uint32_t n_rows, double identity, int rescue_small, uint32_t** pair_i,
                    uint32_t** pair_j, uint64_t* n_pairs) {
    if (!ctx || !set || !pair_i || !pair_j || !n_pairs) return SKH_ERR_INVALID;
    *pair_i = *pair_j = nullptr; *n_pairs = 0;
    int rc = guarded(ctx, [&] {
        if (row0 > set->n_genomes || n_rows > set->n_genomes - row0) throw std::invalid_argument("row range outside the sketch set");
        std::vector<uint32_t> a, b;
        { Stopwatch sw(ctx, &ctx->timings.screen_ms); screen_pairs(ctx, set, nullptr, identity, SKH_SCREEN_REFS, rescue_small, a, b, row0, row0 + n_rows); }
        uint32_t* pa = (uint32_t*)malloc((a.size() + 1) * 4); uint32_t* pb = (uint32_t*)malloc((b.size() + 1) * 4);
        if (!pa || !pb) { free(pa); free(pb); throw std::bad_alloc(); }
        memcpy(pa, a.data(), a.size() * 4); memcpy(pb, b.data(), b.size() * 4);
        *pair_i = pa; *pair_j = pb; *n_pairs = a.size();
    });
    ctx->arena.reset();
    return rc;
}

int skh_chain_pairs(skh_ctx* ctx, const skh_sketch_set* refs, const skh_sketch_set* queries, const uint32_t* pair_ref, const uint32_t* pair_query,
                    uint64_t n_pairs, const skh_map_params* mp, skh_ani_result* out, skh_chain_stats* stats) {
    if (!ctx || !refs || !mp || (n_pairs && (!pair_ref || !pair_query || !out))) return SKH_ERR_INVALID;
    int rc = guarded(ctx, [&] {
        Stopwatch sw(ctx, &ctx->timings.chain_ms);
        const skh_sketch_set* q = queries ? queries : refs;
        chain_pairs(ctx, &refs, 1, nullptr, &q, 1, nullptr, pair_ref, pair_query, n_pairs, *mp, out, stats);
    });
    ctx->arena.reset();
    return rc;
}

static void out_u32(const std::vector<uint32_t>& v, uint32_t** out) {
    uint32_t* p = (uint32_t*)malloc((v.size() + 1) * 4);
    if (!p) throw std::bad_alloc();
    if (!v.empty()) memcpy(p, v.data(), v.size() * 4);
    *out = p;
}

int skh_screen_part(skh_ctx* ctx, const skh_sketch_set* set, uint32_t part, uint32_t n_parts, uint64_t** cells, uint64_t* n_cells) {
    if (!ctx || !set || !cells || !n_cells || !n_parts || part >= n_parts) return SKH_ERR_INVALID;
    *cells = nullptr; *n_cells = 0;
    int rc = guarded(ctx, [&] {
        if (!screen_parts_fit(ctx, set->n_genomes) && set->n_genomes) throw std::invalid_argument("the count matrix of this set is beyond the screen's budget");
        std::vector<uint64_t> c;
        { Stopwatch sw(ctx, &ctx->timings.screen_ms); if (set->n_genomes) screen_partial_cells(ctx, set, part, n_parts, c); }
        uint64_t* p = (uint64_t*)malloc((c.size() + 1) * 8);
        if (!p) throw std::bad_alloc();
        if (!c.empty()) memcpy(p, c.data(), c.size() * 8);
        *cells = p; *n_cells = c.size();
    });
    ctx->arena.reset();
    return rc;
}

int skh_screen_from_cells(skh_ctx* ctx, const skh_sketch_set* set, const uint64_t* cells, uint64_t n_cells, double identity, int rescue_small,
                          uint32_t** pair_i, uint32_t** pair_j, uint64_t* n_pairs) {
    if (!ctx || !set || !pair_i || !pair_j || !n_pairs || (n_cells && !cells)) return SKH_ERR_INVALID;
    *pair_i = *pair_j = nullptr; *n_pairs = 0;
    int rc = guarded(ctx, [&] {
        if (!screen_parts_fit(ctx, set->n_genomes) && set->n_genomes) throw std::invalid_argument("the count matrix of this set is beyond the screen's budget");
        std::vector<uint32_t> a, b;
        { Stopwatch sw(ctx, &ctx->timings.screen_ms); if (set->n_genomes) screen_from_cells(ctx, set, cells, n_cells, identity, rescue_small, a, b); }
        out_u32(a, pair_i); out_u32(b, pair_j); *n_pairs = a.size();
    });
    if (rc != SKH_OK) { free(*pair_i); free(*pair_j); *pair_i = *pair_j = nullptr; }
    ctx->arena.reset();
    return rc;
}

int skh_chain_pairs_multi(skh_ctx* ctx, const skh_sketch_set* const* ref_sets, uint32_t n_ref_sets, const skh_sketch_set* queries, const uint32_t* pair_set,
                          const uint32_t* pair_ref, const uint32_t* pair_query, uint64_t n_pairs, const skh_map_params* mp, skh_ani_result* out) {
    if (!ctx || !ref_sets || !n_ref_sets || !queries || !mp || (n_pairs && (!pair_set || !pair_ref || !pair_query || !out))) return SKH_ERR_INVALID;
    int rc = guarded(ctx, [&] {
        Stopwatch sw(ctx, &ctx->timings.chain_ms);
        chain_pairs(ctx, ref_sets, n_ref_sets, pair_set, &queries, 1, nullptr, pair_ref, pair_query, n_pairs, *mp, out, nullptr);
    });
    ctx->arena.reset();
    return rc;
}

int skh_triangle(skh_ctx* ctx, const skh_sketch_set* ss, double identity, int rescue_small, const skh_map_params* mp, uint32_t part,
                 uint32_t n_parts, uint32_t** out_i, uint32_t** out_j, skh_ani_result** out_res, uint64_t* n_kept, uint64_t* n_chained) {
    if (!ctx || !ss || !mp || !out_i || !out_j || !out_res || !n_kept || n_parts == 0 || part >= n_parts) return SKH_ERR_INVALID;
    *out_i = *out_j = nullptr; *out_res = nullptr; *n_kept = 0;
    StageTrace tr_entry(ctx);
    int rc = guarded(ctx, [&] {
        tr_entry.mark("triangle: entry");
        StageTrace tr(ctx);
        std::vector<uint32_t> a, b;
        // A set sketched with deferred tables: its seed tables are queued on the main stream NOW, and the screen (marker incidences, count matrix, pair
        // list read-back) and the host's pair descriptors are made beside them on the second stream -- the screen needs the marker sets only.
        skh_sketch_set* ssm = const_cast<skh_sketch_set*>(ss);
        std::unique_lock<std::mutex> build_lock(ssm->build_mu, std::defer_lock);
        bool overlapped = false; TableBuild tb; DevEvent ev_b0, ev_b1;
        if (!ss->tables_built) {
            build_lock.lock();
            if (!ss->tables_built) { overlapped = true; ev_b0.record(ctx->stream); tb = build_sketch_tables_begin(ctx, ssm, nullptr, nullptr); ev_b1.record(ctx->stream); }
            else build_lock.unlock();
        }
        struct BuildGuard { bool* armed; ~BuildGuard() { if (*armed) device_sync_all(); } } build_guard{&overlapped};   // queued kernels never outlive the arena on an error
        if (overlapped) {
            std::swap(ctx->stream, ctx->stream2);
            try { Stopwatch sw(ctx, &ctx->timings.screen_ms); screen_pairs(ctx, ss, nullptr, identity, SKH_SCREEN_REFS, rescue_small, a, b); }
            catch (...) { std::swap(ctx->stream, ctx->stream2); throw; }
            std::swap(ctx->stream, ctx->stream2);
        } else {
            { Stopwatch sw(ctx, &ctx->timings.screen_ms); screen_pairs(ctx, ss, nullptr, identity, SKH_SCREEN_REFS, rescue_small, a, b); }
            ctx->arena.reset();
        }
        tr.mark("triangle: screen");
        std::vector<uint32_t> pi, pj;
        if (n_parts == 1) { pi.swap(a); pj.swap(b); }
        else for (size_t p = part; p < a.size(); p += n_parts) { pi.push_back(a[p]); pj.push_back(b[p]); }   // triangle.rs:89-98: ref = i, query = j
        const size_t n_res = pi.size();
        skh_ani_result* res = (skh_ani_result*)ctx->pin_results.need((n_res + 1) * sizeof(skh_ani_result));   // (every row is written by the chaining or the call fails)
        tr.mark("triangle: pair list");
        const std::function<void()> finish_tables = [&] {
            build_sketch_tables_finish(ctx, ssm, tb); overlapped = false;
            ctx->timings.sketch_build_ms += DevEvent::ms(ev_b0, ev_b1);                // (the build's kernels as the main stream ran them, the screen beside them)
            build_lock.unlock();
        };
        { Stopwatch sw(ctx, &ctx->timings.chain_ms);
          chain_pairs(ctx, &ss, 1, nullptr, &ss, 1, nullptr, pi.data(), pj.data(), pi.size(), *mp, res, nullptr, false, overlapped ? &finish_tables : nullptr); }
        tr.mark("triangle: chain");
        if (n_chained) *n_chained = pi.size();
        size_t kept = 0;
        for (size_t p = 0; p < n_res; p++) if (res[p].ani > 0.1f) kept++;                                  // triangle.rs:99
        uint32_t* oi = (uint32_t*)malloc((kept + 1) * 4); uint32_t* oj = (uint32_t*)malloc((kept + 1) * 4);
        skh_ani_result* orr = (skh_ani_result*)malloc((kept + 1) * sizeof(skh_ani_result));
        if (!oi || !oj || !orr) { free(oi); free(oj); free(orr); throw std::bad_alloc(); }
        size_t q = 0;
        for (size_t p = 0; p < n_res; p++) if (res[p].ani > 0.1f) { oi[q] = pi[p]; oj[q] = pj[p]; orr[q] = res[p]; q++; }
        *out_i = oi; *out_j = oj; *out_res = orr; *n_kept = kept;
        tr.mark("triangle: results out");
    });
    StageTrace tr_exit(ctx);
    ctx->arena.reset();
    tr_exit.mark("triangle: arena reset");
    return rc;
}

int skh_comm_create_host(skh_ctx* ctx, const skh_host_collectives* hc, int rank, int world, skh_comm** out) {
    if (!ctx || !hc || !out) return SKH_ERR_INVALID;
    *out = nullptr;
    return guarded(ctx, [&] {
        if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("bad rank / world size");
        skh_comm* c = new skh_comm(); c->t = make_host_transport(hc, rank, world); *out = c;
    });
}

void skh_comm_destroy(skh_comm* c) { delete c; }

int skh_comm_selftest(skh_ctx* ctx, skh_comm* c) {
    if (!ctx || !c || !c->t) return SKH_ERR_INVALID;
    int rc = guarded(ctx, [&] { comm_selftest(ctx, *c->t); });
    ctx->arena.reset();
    return rc;
}

int skh_plan_pairs(uint32_t n_genomes, const uint32_t* pair_i, const uint32_t* pair_j, uint64_t n_pairs, const uint64_t* weight, const uint32_t* holder, int world,
                   uint8_t* owner) {
    if ((n_pairs && (!pair_i || !pair_j || !owner)) || !weight || world < 1 || world > 255) return SKH_ERR_INVALID;
    try {
        for (uint64_t p = 0; p < n_pairs; p++) if (pair_i[p] >= n_genomes || pair_j[p] >= n_genomes) return SKH_ERR_INVALID;
        std::vector<uint32_t> pi(pair_i, pair_i + n_pairs), pj(pair_j, pair_j + n_pairs); std::vector<uint64_t> w(weight, weight + n_genomes), units, load;
        std::vector<uint8_t> own; std::vector<int> hold;
        if (holder) { hold.resize(n_genomes); for (uint32_t g = 0; g < n_genomes; g++) { if (holder[g] >= (uint32_t)world) return SKH_ERR_INVALID; hold[g] = (int)holder[g]; } }
        assign_pairs(n_genomes, pi, pj, w, hold, world, own, units, load);
        if (n_pairs) memcpy(owner, own.data(), n_pairs);
        return SKH_OK;
    } catch (...) { return SKH_ERR_INTERNAL; }
}

int skh_triangle_distributed(skh_ctx* ctx, skh_comm* comm, const skh_sketch_set* local, double identity, int rescue_small, const skh_map_params* mp,
                             uint32_t** out_i, uint32_t** out_j, skh_ani_result** out_res, uint64_t* n_kept, uint64_t* n_chained, skh_dist_stats* stats) {
    return skh_triangle_distributed_ex(ctx, comm, local, identity, rescue_small, mp, 0, out_i, out_j, out_res, n_kept, n_chained, stats);
}

int skh_triangle_distributed_ex(skh_ctx* ctx, skh_comm* comm, const skh_sketch_set* local, double identity, int rescue_small, const skh_map_params* mp, uint32_t flags,
                                uint32_t** out_i, uint32_t** out_j, skh_ani_result** out_res, uint64_t* n_kept, uint64_t* n_chained, skh_dist_stats* stats) {
    if (!ctx || !comm || !comm->t || !local || !mp || !out_i || !out_j || !out_res || !n_kept) return SKH_ERR_INVALID;
    *out_i = *out_j = nullptr; *out_res = nullptr; *n_kept = 0;
    int rc = guarded(ctx, [&] {
        std::vector<uint32_t> a, b; std::vector<skh_ani_result> r;
        triangle_distributed(ctx, *comm->t, local, identity, rescue_small, *mp, flags, a, b, r, n_chained, stats);
        const size_t kept = a.size();
        uint32_t* oi = (uint32_t*)malloc((kept + 1) * 4); uint32_t* oj = (uint32_t*)malloc((kept + 1) * 4);
        skh_ani_result* orr = (skh_ani_result*)malloc((kept + 1) * sizeof(skh_ani_result));
        if (!oi || !oj || !orr) { free(oi); free(oj); free(orr); throw std::bad_alloc(); }
        memcpy(oi, a.data(), kept * 4); memcpy(oj, b.data(), kept * 4); memcpy(orr, r.data(), kept * sizeof(skh_ani_result));
        *out_i = oi; *out_j = oj; *out_res = orr; *n_kept = kept;
    });
    ctx->arena.reset();
    return rc;
}

int skh_get_timings(const skh_ctx* ctx, skh_timings* out) {
    if (!ctx || !out) return SKH_ERR_INVALID;
    *out = ctx->timings;
    const_cast<skh_ctx*>(ctx)->timings = skh_timings{};
    return SKH_OK;
}

}  // extern "C"
