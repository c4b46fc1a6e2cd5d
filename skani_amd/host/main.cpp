// main.cpp -- `skani-hip {triangle,dist,sketch,search}`: the reference's drivers (triangle.rs:13-169, dist.rs:12-190,
// sketch.rs:15-175, search.rs:16-282) over the C ABI.  Inputs may be FASTA (.gz) or skani sketch files / database folders.
// Flag names follow cli.rs; only flags that reach the hot path or the writers are implemented.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "host.hpp"

using namespace skhost;

namespace {

struct Args {
    std::string cmd, out, list, qlist, rlist, models_dir, database;
    std::vector<std::string> files, queries, refs;
    uint32_t c = 125, k = 15, m = 1000; bool c_set = false, m_set = false;
    double s = 0, min_af = -1, both_min_af = -1; bool min_af_set = false;
    bool robust = false, median = false, no_learned = false, fast = false, slow = false, medium = false, small_genomes = false, faster_small = false;
    bool sparse = false, individual = false, qi = false, ri = false, marker_index = false, no_marker_index = false, separate_sketches = false;
    size_t n = 10000000; int threads = 3, device = 0, seeding_mode = SKH_SEED_AVX2;
    int gpus = 1; bool one_device = false;                                        // triangle --gpus N: one process per GPU (--one-device: all ranks on --device, host collectives)
    uint64_t shard_positions = 1500000000ull;                                     // seed positions per resident database shard (search)
    OutOpts o;
};

// A rank of `triangle --gpus N` (g_node set) that fails ends the whole run: the first rank to claim the failure prints its message -- the run's ONE error --, the
// others leave quietly (their collectives fail once the flag is up; a rank that only learnt of a peer's failure waits a moment so that the peer's own message wins).
Node* g_node = nullptr;
[[noreturn]] void die(const std::string& m, bool secondary = false) {               // secondary: this rank only learnt that a PEER failed (SKH_ERR_PEER, a node collective cut short)
    if (!g_node) { fprintf(stderr, "ERROR %s\n", m.c_str()); exit(1); }
    for (int w = 0; secondary && w < 300 && !node_failed(g_node); w++) usleep(1000);  // the peer's own message is the run's error: wait (a moment at most) until it has claimed it
    if (node_claim_failure(g_node)) fprintf(stderr, "ERROR [rank %d] %s\n", node_rank(g_node), m.c_str());
    fflush(stderr); fflush(stdout);
    _exit(1);
}

std::vector<std::string> read_list(const std::string& p) {
    std::ifstream f(p); if (!f) die("cannot read list file " + p);
    std::vector<std::string> v; std::string l;
    while (std::getline(f, l)) { while (!l.empty() && (l.back() == '\r' || l.back() == ' ')) l.pop_back(); if (!l.empty()) v.push_back(l); }
    return v;
}

Args parse(int argc, char** argv) {
    Args a;
    if (argc < 2) die("usage: skani-hip {triangle|dist|sketch|search} [options] files...");
    a.cmd = argv[1];
    auto need = [&](int& i) -> std::string { if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]); return argv[++i]; };
    std::vector<std::string>* sink = &a.files;
    for (int i = 2; i < argc; i++) {
        std::string x = argv[i];
        if (x == "-o") { a.out = need(i); sink = &a.files; }
        else if (x == "-l") { a.list = need(i); }
        else if (x == "--ql") a.qlist = need(i);
        else if (x == "--rl") a.rlist = need(i);
        else if (x == "-q") sink = &a.queries;
        else if (x == "-r") sink = &a.refs;
        else if (x == "-c") { a.c = (uint32_t)atoi(need(i).c_str()); a.c_set = true; }
        else if (x == "-k") a.k = (uint32_t)atoi(need(i).c_str());
        else if (x == "-m") { a.m = (uint32_t)atoi(need(i).c_str()); a.m_set = true; }
        else if (x == "-s") a.s = atof(need(i).c_str());
        else if (x == "-n") a.n = (size_t)atoll(need(i).c_str());
        else if (x == "-t") a.threads = atoi(need(i).c_str());
        else if (x == "--min-af") { a.min_af = atof(need(i).c_str()); a.min_af_set = true; }
        else if (x == "--both-min-af") a.both_min_af = atof(need(i).c_str());
        else if (x == "--robust") a.robust = true;
        else if (x == "--median") a.median = true;
        else if (x == "--no-learned-ani") a.no_learned = true;
        else if (x == "--fast") a.fast = true;
        else if (x == "--slow") a.slow = true;
        else if (x == "--medium") a.medium = true;
        else if (x == "--small-genomes") a.small_genomes = true;
        else if (x == "--faster-small") a.faster_small = true;
        else if (x == "-E" || x == "--sparse") a.sparse = true;
        else if (x == "--full-matrix") a.o.full_matrix = true;
        else if (x == "--diagonal") a.o.diagonal = true;
        else if (x == "--distance") a.o.distance = true;
        else if (x == "--ci") a.o.ci = true;
        else if (x == "--detailed") a.o.detailed = true;
        else if (x == "--short-header") a.o.short_header = true;
        else if (x == "-i") a.individual = true;
        else if (x == "--qi") a.qi = true;
        else if (x == "--ri") a.ri = true;
        else if (x == "--marker-index") a.marker_index = true;
        else if (x == "--no-marker-index") a.no_marker_index = true;
        else if (x == "--separate-sketches") a.separate_sketches = true;
        else if (x == "-d") a.database = need(i);
        else if (x == "--shard-positions") a.shard_positions = (uint64_t)atoll(need(i).c_str());
        else if (x == "--keep-refs") {}                                        // every reference sketch is HBM-resident anyway
        else if (x == "--device") a.device = atoi(need(i).c_str());
        else if (x == "--gpus") a.gpus = atoi(need(i).c_str());
        else if (x == "--one-device") a.one_device = true;
        else if (x == "--seeding") { std::string v = need(i); a.seeding_mode = v == "scalar" ? SKH_SEED_SCALAR : SKH_SEED_AVX2; }
        else if (x == "--models") a.models_dir = need(i);
        else if (x == "-v" || x == "--debug" || x == "--trace") {}
        else if (!x.empty() && x[0] == '-') die("unknown option " + x);
        else sink->push_back(x);
    }
    // presets (parse.rs:820-853)
    if (a.fast && a.slow) die("Both --slow and --fast were set. This is not allowed.");
    if (a.fast) a.c = 200;
    if (a.slow) a.c = 30;
    if (a.medium) a.c = 70;
    if (a.small_genomes) { a.c = 30; a.m = 200; }
    if (a.c > a.m) die("We currently don't allow c > m. -m should be larger than c.");     // params.rs:183-185
    return a;
}

struct Ctx {
    skh_ctx* c = nullptr;
    void check(int rc, const char* what) { if (rc != 0) die(std::string(what) + ": " + (c ? skh_last_error(c) : "no context"), rc == SKH_ERR_PEER); }
};

std::string models_dir(const Args& a, const char* argv0) {
    if (!a.models_dir.empty()) return a.models_dir;
    if (const char* e = getenv("SKANI_HIP_DATA")) return e;
    std::string p = argv0; size_t s = p.rfind('/');
    return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/../data";
}

skh_sketch_set* sketch(Ctx& cx, const LoadedGenomes& lg, const Args& a, const uint32_t* genome_rank = nullptr) {
    skh_sketch_params sp{a.c, a.k, a.m, (uint32_t)a.seeding_mode};
    skh_sketch_set* ss = nullptr;
    // genome order is already the sorted file-name order => genome_rank = index (chain.rs:20-22 tie rule); a rank of several passes its genomes' places in the whole list
    cx.check(skh_sketch_batch(cx.c, (const uint8_t*)lg.bases.data(), lg.contig_off.data(), lg.contig_genome.data(), (uint32_t)lg.contig_genome.size(),
                              (uint32_t)lg.info.size(), &sp, genome_rank, &ss), "skh_sketch_batch");
    std::vector<const char*> names; for (auto& g : lg.info) names.push_back(g.file_name.c_str());
    cx.check(skh_sketch_set_names(ss, names.data()), "skh_sketch_set_names");      // exact switch_qr tie-break across ref/query sets
    return ss;
}

// SKH_TIMING=1: wall-clock of the driver's phases on stderr, one JSON object (bench.py --workload e2e reads it)
struct PhaseClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0; bool on = getenv("SKH_TIMING") != nullptr;
    std::vector<std::pair<std::string, double>> phases;                              // a phase that is marked twice (the two sides of `dist`) adds up
    void mark(const char* name) {
        const auto n = std::chrono::steady_clock::now(); const double dt = std::chrono::duration<double>(n - last).count(); last = n;
        for (auto& p : phases) if (p.first == name) { p.second += dt; return; }
        phases.emplace_back(name, dt);
    }
    void done() {
        if (!on) return;
        std::string json;
        for (auto& p : phases) { char b[96]; snprintf(b, sizeof b, "\"%s_s\": %.6f, ", p.first.c_str(), p.second); json += b; }
        fprintf(stderr, "{%s\"total_s\": %.6f}\n", json.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};
PhaseClock g_clock;

// fastx_to_sketches (file_io.rs:141-252) with parsing, PCIe and packing overlapped.  The reference parses its files in parallel (file_io.rs:147) and sketches
// each on the thread that read it; here `threads` parser threads write the kept contigs of the files (mapped, copied once) into a few pinned buffers and
// the thread that completes a buffer hands it to the GPU (skh_genomes_append: asynchronous copy + pack kernel) while the others keep parsing.  The
// files are laid out beforehand: consecutive files of the sorted list share a buffer, every file gets a stretch of its own size, the buffers rotate
// through SLOTS pinned allocations.  Genome number = the file's position in the sorted list (file_io.rs:250); files that end up without a kept
// contig are taken out of the numbering afterwards.  Only for plain FASTA files; anything gzipped or FASTQ sends the whole run through load_genomes().
// The pinned buffers of the streaming ingest outlive a call: `dist` runs it once per side, and page-locking a buffer the size of a chromosome file takes
// longer than parsing the file (0.45 s for 2.3 GB).  Handed back by main() at the end.
struct PinnedPool {
    std::vector<std::pair<uint8_t*, uint64_t>> have;
    uint8_t* take(uint64_t bytes) {
        for (auto& b : have) if (b.first && b.second >= bytes) { uint8_t* p = b.first; b.first = nullptr; return p; }
        for (auto& b : have) if (b.first) { skh_host_free(b.first); b.first = nullptr; }                  // too small: make room before asking for more
        return (uint8_t*)skh_host_alloc(bytes);
    }
    void give(uint8_t* p, uint64_t bytes) { if (!p) return; for (auto& b : have) if (!b.first) { b = {p, bytes}; return; } have.emplace_back(p, bytes); }
    void release() { for (auto& b : have) if (b.first) { skh_host_free(b.first); b.first = nullptr; } }   // (called by main before the runtime goes away)
};
PinnedPool g_pinned;

struct Streamed { bool ok = false; skh_sketch_set* ss = nullptr; std::vector<GenomeInfo> info; std::vector<uint32_t> kept_index; };   // kept_index[set genome] = index into info, or ~0u
// rank_base / sketch_flags: a rank of `triangle --gpus N` streams its share of the sorted list -- genome_rank = rank_base + the file's place in the share, the seed
// tables deferred (skh_triangle_distributed indexes only what the rank ends up chaining)
Streamed stream_side(Ctx& cx, const std::vector<std::string>& files_in, const Args& a, uint32_t rank_base = 0, uint32_t sketch_flags = 0) {
    Streamed out;
    std::vector<std::string> files = files_in;
    std::stable_sort(files.begin(), files.end());
    const uint32_t nf = (uint32_t)files.size();
    uint64_t total = 0, largest = 0; std::vector<uint64_t> fsize(nf);
    for (uint32_t i = 0; i < nf; i++) {
        struct stat sb;
        if (stat(files[i].c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return out;
        FILE* f = fopen(files[i].c_str(), "rb"); if (!f) return out;
        unsigned char m[2] = {0, 0}; const size_t got = fread(m, 1, 2, f); fclose(f);
        if (got == 2 && m[0] == 0x1f && m[1] == 0x8b) return out;                  // gzip somewhere: the plain path is not for this run
        fsize[i] = (uint64_t)sb.st_size; total += fsize[i]; largest = std::max(largest, fsize[i]);
    }
    // buffers: consecutive files up to SLOT bytes; buffer b lives in pinned slot b % SLOTS
    const uint64_t SLOT = std::max<uint64_t>((uint64_t)64 << 20, largest + 64);
    const uint32_t SLOTS = (uint32_t)std::max<uint64_t>(2, std::min<uint64_t>(6, ((uint64_t)3 << 29) / SLOT));
    struct Buf { uint32_t f0 = 0, f1 = 0; std::atomic<uint32_t> remaining{0}; };
    std::vector<uint32_t> buf_of(nf); std::vector<uint64_t> off_in(nf);
    std::vector<std::unique_ptr<Buf>> bufs;
    {
        uint64_t at = SLOT + 1;
        for (uint32_t i = 0; i < nf; i++) {
            if (at + fsize[i] + 16 > SLOT) { bufs.emplace_back(new Buf()); bufs.back()->f0 = i; at = 0; }
            buf_of[i] = (uint32_t)bufs.size() - 1; off_in[i] = at; at += (fsize[i] + 15) / 16 * 16;
            bufs.back()->f1 = i + 1; bufs.back()->remaining++;
        }
    }
    std::vector<uint8_t*> slot(SLOTS, nullptr);
    for (uint32_t x = 0; x < std::min<uint32_t>(SLOTS, (uint32_t)bufs.size()); x++) if (!(slot[x] = g_pinned.take(SLOT))) die("cannot pin host memory for the ingest buffers");
    std::vector<uint64_t> slot_gen(SLOTS, 0);                                       // buffers of slot x up to generation slot_gen[x] have been copied: the next may be written
    std::mutex slot_mu; std::condition_variable slot_cv;
    skh_genome_set* gs = nullptr;
    cx.check(skh_genomes_begin(cx.c, total + total / 7 + 4096, nf * 32 + 64, nf, a.seeding_mode, &gs), "skh_genomes_begin");
    std::vector<GenomeInfo> per(nf); std::vector<uint8_t> state(nf, 0);          // 1 = kept, 2 = no kept contig, 3 = unreadable
    std::vector<std::vector<uint64_t>> clens(nf);
    std::atomic<uint32_t> next{0}; std::atomic<bool> fallback{false}; std::mutex gpu; std::string gpu_err;
    const int per_file = std::max(1, a.threads / (int)std::max<uint32_t>(1, nf));      // few files, many threads: the threads of a large file's parse (fastx.cpp)
    auto worker = [&]() {
        std::vector<std::string> names;
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= nf) break;
            Buf& B = *bufs[buf_of[i]];
            const uint32_t x = buf_of[i] % SLOTS; const uint64_t gen = buf_of[i] / SLOTS;
            { std::unique_lock<std::mutex> lk(slot_mu); slot_cv.wait(lk, [&] { return slot_gen[x] >= gen || fallback; }); }
            if (!fallback) {
                size_t wrote = 0;
                try {
                    if (!parse_fasta_plain(files[i], slot[x] + off_in[i], (size_t)fsize[i] + 16, 500, &wrote, names, clens[i], per_file)) { { std::lock_guard<std::mutex> lk(slot_mu); fallback = true; } slot_cv.notify_all(); }
                    else if (names.empty()) state[i] = 2;
                    else { state[i] = 1; GenomeInfo& gi = per[i]; gi.file_name = files[i]; gi.contigs = std::move(names); for (uint64_t l : clens[i]) gi.contig_lengths.push_back((uint32_t)l); }
                } catch (const std::exception& e) { state[i] = 3; fprintf(stderr, "WARN %s; skipping.\n", e.what()); }
            }
            if (B.remaining.fetch_sub(1) != 1) continue;
            // this thread completed the buffer: hand it over, wait for its copy, free the slot for the buffer after next
            if (!fallback) {
                std::vector<uint64_t> starts, lens; std::vector<uint32_t> genome;
                for (uint32_t f = B.f0; f < B.f1; f++) {
                    if (state[f] != 1) continue;
                    uint64_t at = off_in[f];
                    for (uint64_t l : clens[f]) { starts.push_back(at); lens.push_back(l); genome.push_back(f); at += l; }
                }
                if (!starts.empty()) {
                    uint64_t ticket = 0; int rc;
                    { std::lock_guard<std::mutex> lk(gpu);
                      rc = skh_genomes_append(gs, slot[x], starts.data(), lens.data(), genome.data(), (uint32_t)starts.size(), 0, &ticket);
                      if (rc != 0 && gpu_err.empty()) gpu_err = skh_last_error(cx.c); }
                    if (rc == 0 && (rc = skh_genomes_wait(gs, ticket)) != 0) {      // (thread-safe: other threads keep appending)
                        std::lock_guard<std::mutex> lk(gpu); if (gpu_err.empty()) gpu_err = "skh_genomes_wait: the copy of a batch failed on the device";
                    }
                    if (rc != 0) { std::lock_guard<std::mutex> lk(slot_mu); fallback = true; }
                }
            }
            { std::lock_guard<std::mutex> lk(slot_mu); slot_gen[x] = gen + 1; }
            slot_cv.notify_all();
        }
    };
    {
        const int nt = std::max(1, std::min<int>(a.threads, (int)nf));
        std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(worker); for (auto& t : th) t.join();
    }
    if (fallback) {
        (void)skh_genomes_finish(gs);                                              // drains the queued copies before the buffers go away
        skh_genomes_destroy(gs);
        for (uint8_t* p : slot) g_pinned.give(p, SLOT);
        if (!gpu_err.empty()) die("streamed ingest: " + gpu_err);
        return out;                                                                // a FASTQ / gzip file turned up: the caller reads everything the other way
    }
    cx.check(skh_genomes_finish(gs), "skh_genomes_finish");
    for (uint8_t* p : slot) g_pinned.give(p, SLOT);
    g_clock.mark("parse_upload_pack");
    out.kept_index.assign(nf, ~0u);
    for (uint32_t i = 0; i < nf; i++) {
        // (a file without a kept contig stays in the set as a genome WITHOUT CONTIGS under its number in the file list: the screen's small-genome rescue
        // pairs it with every later genome, each such pair is chained as an empty one -- no join tiles, ani 0, dropped from the output -- and is counted in
        // the driver's "chained" statistic; the reference makes no sketch for such a file, file_io.rs:176-214.  Rare, and cheaper than a second numbering.)
        if (state[i] == 1) { out.kept_index[i] = (uint32_t)out.info.size(); out.info.push_back(std::move(per[i])); }
        else if (state[i] == 2) fprintf(stderr, "WARN File %s consists of only contigs < 500 bp. Skipping this file.\n", files[i].c_str());
    }
    skh_sketch_params sp{a.c, a.k, a.m, (uint32_t)a.seeding_mode};
    std::vector<uint32_t> granks(nf); for (uint32_t i = 0; i < nf; i++) granks[i] = rank_base + i;
    cx.check(skh_sketch_genomes_ex(cx.c, gs, &sp, granks.data(), sketch_flags, &out.ss), "skh_sketch_genomes");   // genome_rank = number in the sorted file list
    skh_genomes_destroy(gs);
    std::vector<const char*> nm(nf); for (uint32_t i = 0; i < nf; i++) nm[i] = files[i].c_str();
    if (nf) cx.check(skh_sketch_set_names(out.ss, nm.data()), "skh_sketch_set_names");
    g_clock.mark("sketch");
    out.ok = true;
    return out;
}

bool all_sketch_files(const std::vector<std::string>& files) {                     // parse.rs:264-281
    if (files.empty()) return false;
    for (auto& f : files) if (f.find(".sketch") == std::string::npos && f.find("markers.bin") == std::string::npos) return false;
    return true;
}

// sketches read from disk -> one device-resident sketch set (sketches_from_sketch + the HBM upload)
skh_sketch_set* import_blobs(Ctx& cx, const SketchFileParams& fp, const std::vector<SketchBlob>& blobs, int seeding_mode, const uint32_t* genome_rank = nullptr) {
    std::vector<uint64_t> pos_off{0}, mk_off{0}, ctg_off{0}, total;
    std::vector<uint32_t> seed, pos, cc, clen; std::vector<uint64_t> markers;
    for (const SketchBlob& b : blobs) {
        for (const SeedRecord& r : b.records) { seed.push_back(r.seed); pos.push_back(r.pos); cc.push_back(r.ctgcanon); }
        markers.insert(markers.end(), b.markers.begin(), b.markers.end());
        clen.insert(clen.end(), b.contig_lengths.begin(), b.contig_lengths.end());
        pos_off.push_back(seed.size()); mk_off.push_back(markers.size()); ctg_off.push_back(clen.size()); total.push_back(b.total_sequence_length);
    }
    skh_sketch_params sp{(uint32_t)fp.c, (uint32_t)fp.k, (uint32_t)fp.marker_c, (uint32_t)seeding_mode};
    skh_sketch_set* ss = nullptr;
    cx.check(skh_sketch_import(cx.c, &sp, (uint32_t)blobs.size(), pos_off.data(), seed.data(), pos.data(), cc.data(), mk_off.data(), markers.data(),
                               ctg_off.data(), clen.data(), total.data(), genome_rank, &ss), "skh_sketch_import");
    std::vector<const char*> names; for (auto& b : blobs) names.push_back(b.file_name.c_str());
    cx.check(skh_sketch_set_names(ss, names.data()), "skh_sketch_set_names");
    return ss;
}

std::vector<GenomeInfo> infos_of(const std::vector<SketchBlob>& blobs) {
    std::vector<GenomeInfo> v(blobs.size());
    for (size_t i = 0; i < blobs.size(); i++) { v[i].file_name = blobs[i].file_name; v[i].contigs = blobs[i].contigs; v[i].contig_lengths = blobs[i].contig_lengths; v[i].contig_order = blobs[i].contig_order; }
    return v;
}

// device-resident sketch set -> the records skani serialises (Sketch, types.rs:252-277)
std::vector<SketchBlob> export_blobs(Ctx& cx, const skh_sketch_set* ss, const std::vector<GenomeInfo>& info, const Args& a) {
    std::vector<SketchBlob> blobs(info.size());
    for (uint32_t g = 0; g < info.size(); g++) {
        uint64_t np = 0, nd = 0, nm = 0, total = 0; uint32_t nc = 0;
        cx.check(skh_sketch_sizes(ss, g, &np, &nd, &nm, &nc, &total), "skh_sketch_sizes");
        std::vector<uint32_t> seed(np), pos(np), cc(np), clen(nc); SketchBlob& b = blobs[g];
        b.markers.resize(nm);
        cx.check(skh_sketch_export(ss, g, seed.data(), pos.data(), cc.data(), b.markers.data(), clen.data()), "skh_sketch_export");
        b.records.resize(np); for (uint64_t i = 0; i < np; i++) b.records[i] = SeedRecord{seed[i], pos[i], cc[i]};
        b.file_name = info[g].file_name; b.contigs = info[g].contigs; b.contig_lengths = clen; b.total_sequence_length = total;
        b.marker_c = a.c;                                                          // Sketch::new stores c in marker_c (types.rs:346)
        b.c = a.c; b.k = a.k; b.contig_order = info[g].contig_order;
    }
    return blobs;
}

// One side of a comparison: FASTA files are sketched on the GPU, sketch files are uploaded.  `a.c/k/m` are replaced by the
// files' parameters when sketches are read (dist.rs:28-52 "Using parameters from .sketch files").
struct Side { skh_sketch_set* ss = nullptr; std::vector<GenomeInfo> info; bool from_sketch = false; };
Side load_side(Ctx& cx, const std::vector<std::string>& files, bool individual, Args& a) {
    Side sd;
    if (all_sketch_files(files)) {
        SketchFileParams fp; std::vector<SketchBlob> blobs = read_sketch_files(files, fp);
        if (blobs.empty()) return sd;
        for (auto& b : blobs) if (!b.has_seeds) die("sketch " + b.file_name + " holds markers only; it cannot be aligned");
        a.c = (uint32_t)fp.c; a.k = (uint32_t)fp.k; a.m = (uint32_t)fp.marker_c;
        sd.ss = import_blobs(cx, fp, blobs, a.seeding_mode); sd.info = infos_of(blobs); sd.from_sketch = true;
    } else {
        LoadedGenomes lg = load_genomes(files, individual, a.threads);
        if (lg.info.empty()) return sd;
        sd.ss = sketch(cx, lg, a); sd.info = std::move(lg.info);
    }
    return sd;
}

void emit(const std::string& path, const std::string& text) {
    if (path.empty()) { fwrite(text.data(), 1, text.size(), stdout); return; }
    FILE* f = fopen(path.c_str(), "wb"); if (!f) die("cannot write " + path);
    fwrite(text.data(), 1, text.size(), f); fclose(f);
}

skh_map_params map_params(const Args& a, bool learned) {
    skh_map_params mp{};
    mp.min_af = a.min_af_set ? a.min_af / 100. : 0.15;                       // D_FRAC_COVER_CUTOFF (parse.rs:860-862)
    mp.both_min_af = a.both_min_af / 100.;                                    // default -1/100 = disabled
    mp.robust = a.robust; mp.median = a.median; mp.learned_ani = learned; mp.compute_ci = 1;
    return mp;
}

int run_triangle(Args& a, Ctx& cx) {
    std::vector<std::string> files = a.files;
    if (!a.list.empty()) { auto l = read_list(a.list); files.insert(files.end(), l.begin(), l.end()); }
    if (files.empty()) die("No reference inputs found.");
    Side sd; std::vector<uint32_t> kept_index;                                  // streamed ingest: set genome -> row of the output (files without a kept contig have none)
    if (!a.individual && !all_sketch_files(files)) {
        Streamed st = stream_side(cx, files, a);
        if (st.ok) { sd.ss = st.ss; sd.info = std::move(st.info); kept_index = std::move(st.kept_index); }
    }
    if (!sd.ss) { sd = load_side(cx, files, a.individual, a); g_clock.mark("load_sketch"); }
    if (sd.info.empty()) die("No genomes/sketches found.");                     // triangle.rs:46-49
    skh_sketch_set* ss = sd.ss; struct { std::vector<GenomeInfo>& info; } lg{sd.info};
    const bool learned = !a.no_learned && a.c >= 70 && !a.individual && !a.median;   // regression.rs:8-10, parse.rs:885-889
    skh_map_params mp = map_params(a, learned);
    const bool rescue_small = !a.faster_small && !a.small_genomes;              // parse.rs:798
    uint32_t *oi = nullptr, *oj = nullptr; skh_ani_result* res = nullptr; uint64_t kept = 0, chained = 0;
    cx.check(skh_triangle(cx.c, ss, a.s / 100., rescue_small, &mp, 0, 1, &oi, &oj, &res, &kept, &chained), "skh_triangle");
    g_clock.mark("triangle");
    std::vector<PairResult> pr(kept);
    for (uint64_t x = 0; x < kept; x++) pr[x] = PairResult{kept_index.empty() ? oi[x] : kept_index[oi[x]], kept_index.empty() ? oj[x] : kept_index[oj[x]], res[x]};    // ref = i, query = j (triangle.rs:98)
    skh_free(oi); skh_free(oj); skh_free(res);
    if (a.sparse) emit(a.out, format_sparse(lg.info, pr, a.o));
    else {
        std::string ani, af; format_phylip(lg.info, pr, a.individual, a.o, ani, af);
        emit(a.out, ani); emit(a.out.empty() ? std::string("skani_matrix.af") : a.out + ".af", af);   // file_io.rs:428,467
    }
    g_clock.mark("write");
    skh_sketch_set_destroy(ss);
    return 0;
}

// ---- `triangle --gpus N` (triangle.rs:13-169 on the GPUs of one node; node.cpp has the launcher and the ranks' shared-memory collectives).
// Rank r ingests and sketches a contiguous share of the name-sorted file list (file_io.rs:250: the sorted list IS the genome order, so global genome = genomes on
// lower ranks + local genome, which is skh_triangle_distributed's numbering), with deferred seed tables; the library does the rest below the C ABI; the result rows
// come to rank 0 (SKH_DIST_ROWS_TO_ROOT), which writes the matrix / edge list through the same writers as the one-process run.
void put_u64(std::string& b, uint64_t v) { b.append((const char*)&v, 8); }
void put_str(std::string& b, const std::string& x) { put_u64(b, x.size()); b += x; }
struct Reader {
    const std::string& b; size_t at = 0;
    uint64_t u64() { if (at + 8 > b.size()) die("a rank's genome table is cut short"); uint64_t v; memcpy(&v, b.data() + at, 8); at += 8; return v; }
    std::string str() { const uint64_t n = u64(); if (at + n > b.size()) die("a rank's genome table is cut short"); std::string x = b.substr(at, n); at += n; return x; }
};

int run_triangle_node(Args& a, const char* argv0) {
    std::vector<std::string> files = a.files;
    if (!a.list.empty()) { auto l = read_list(a.list); files.insert(files.end(), l.begin(), l.end()); }
    if (files.empty()) die("No reference inputs found.");
    const int W = a.gpus;
    const bool sketch_inputs = all_sketch_files(files);
    std::stable_sort(files.begin(), files.end());
    // shares of the sorted list, cut by bytes (a share = consecutive files; sketch files are read by every rank and cut by count afterwards)
    std::vector<uint32_t> cut(W + 1, 0);
    {
        std::vector<uint64_t> cum(files.size() + 1, 0);
        for (size_t i = 0; i < files.size(); i++) { struct stat sb; cum[i + 1] = cum[i] + (stat(files[i].c_str(), &sb) == 0 ? (uint64_t)sb.st_size : 1) + 1; }
        for (int r = 1; r < W; r++) {
            const uint64_t want = cum.back() / (uint64_t)W * (uint64_t)r;
            cut[r] = std::max<uint32_t>(cut[r - 1], (uint32_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin()));
        }
        cut[W] = (uint32_t)files.size();
    }
    Node* node = node_create(W);
    const int rank = node_launch(node);                                           // (the launcher stays in there)
    g_node = node;
    g_clock = PhaseClock();
    a.threads = std::max(1, a.threads / W);
    if (const char* fr = getenv("SKH_TUNE_DIST_FAIL_RANK")) if (atoi(fr) != rank) unsetenv("SKH_TUNE_DIST_FAIL");   // (fault injection of the tests: SKH_TUNE_DIST_FAIL on one rank only)
    Ctx cx;
    if (skh_ctx_create(a.one_device ? a.device : a.device + rank, &cx.c) != 0) die("no usable MI355X device " + std::to_string(a.one_device ? a.device : a.device + rank) + " (skani-hip has no CPU path)");
    const std::string md = models_dir(a, argv0);
    cx.check(skh_load_models(cx.c, (md + "/gbdt_c125.bin").c_str(), (md + "/gbdt_c200.bin").c_str()), "skh_load_models");
    // ---- the communicator: RCCL inside the library (rank 0's unique id goes round through shared memory), tried out by its self-test; all ranks together fall back to
    // host collectives over shared memory when any of them could not use it, or when they share one device
    skh_comm* comm = nullptr;
    if (!a.one_device && !getenv("SKANI_HIP_HOST_COLLECTIVES")) {
        struct IdMsg { uint8_t id[SKH_COMM_ID_BYTES]; uint64_t ok; } mine{}; std::vector<IdMsg> all(W);
        if (rank == 0) mine.ok = skh_comm_unique_id(mine.id) == 0;
        if (!node_all_gather(node, &mine, all.data(), sizeof mine)) die("a rank failed; all ranks stop", true);
        uint64_t bad = 1;
        if (all[0].ok) {
            bad = skh_comm_create_rccl(cx.c, all[0].id, rank, W, &comm) != 0;
            if (!bad) bad = skh_comm_selftest(cx.c, comm) != 0;
        }
        std::vector<uint64_t> bads(W);
        if (!node_all_gather(node, &bad, bads.data(), 8)) die("a rank failed; all ranks stop", true);
        bool any = false; for (uint64_t b : bads) any = any || b;
        if (any) {
            if (comm) { skh_comm_destroy(comm); comm = nullptr; }
            if (rank == 0) fprintf(stderr, "WARN RCCL is not usable between the %d ranks (%s); the ranks exchange through host memory instead.\n", W, all[0].ok ? "communicator self-test" : "librccl");
        }
    }
    skh_host_collectives hc = node_collectives(node);
    if (!comm) cx.check(skh_comm_create_host(cx.c, &hc, rank, W, &comm), "skh_comm_create_host");
    g_clock.mark("startup");
    // ---- this rank's genomes
    const uint32_t f0 = cut[rank], f1 = cut[rank + 1];
    Side sd; std::vector<uint32_t> kept_index;                                    // local set genome -> local row of `info` (streamed ingest), else the identity
    const uint32_t flags = SKH_SKETCH_DEFER_TABLES | SKH_SKETCH_NO_SCREEN_INDEX;
    if (sketch_inputs) {
        SketchFileParams fp; std::vector<SketchBlob> blobs = read_sketch_files(files, fp);   // (sorted by the name inside the sketch: every rank reads all, keeps its stretch)
        for (auto& b : blobs) if (!b.has_seeds) die("sketch " + b.file_name + " holds markers only; it cannot be aligned");
        a.c = (uint32_t)fp.c; a.k = (uint32_t)fp.k; a.m = (uint32_t)fp.marker_c;
        const size_t b0 = blobs.size() * (size_t)rank / (size_t)W, b1 = blobs.size() * (size_t)(rank + 1) / (size_t)W;
        std::vector<SketchBlob> part(std::make_move_iterator(blobs.begin() + b0), std::make_move_iterator(blobs.begin() + b1));
        std::vector<uint32_t> gr(part.size()); for (size_t x = 0; x < part.size(); x++) gr[x] = (uint32_t)(b0 + x);
        sd.ss = import_blobs(cx, fp, part, a.seeding_mode, gr.data()); sd.info = infos_of(part); sd.from_sketch = true;
        g_clock.mark("load_sketch");
    } else {
        const std::vector<std::string> share(files.begin() + f0, files.begin() + f1);
        if (!a.individual) {
            Streamed st = stream_side(cx, share, a, f0, flags);
            if (st.ok) { sd.ss = st.ss; sd.info = std::move(st.info); kept_index = std::move(st.kept_index); }
        }
        if (!sd.ss) {                                                               // gzip / FASTQ / -i: the record reader
            LoadedGenomes lg = load_genomes(share, a.individual, a.threads);
            std::vector<uint32_t> gr(lg.info.size());                               // the file's place in the whole list (the contigs of a file under -i share it: equal names, no switch)
            for (size_t x = 0; x < lg.info.size(); x++) gr[x] = f0 + (uint32_t)(std::lower_bound(share.begin(), share.end(), lg.info[x].file_name) - share.begin());
            sd.ss = sketch(cx, lg, a, gr.data()); sd.info = std::move(lg.info);
            g_clock.mark("load_sketch");
        }
    }
    const uint32_t n_set = skh_sketch_n_genomes(sd.ss);
    // ---- everybody's genome names to rank 0 (the writers need them); the parameters the ranks ended up with must agree (sketch files carry their own)
    std::string mine_tab; put_u64(mine_tab, a.c); put_u64(mine_tab, a.k); put_u64(mine_tab, a.m); put_u64(mine_tab, n_set); put_u64(mine_tab, sd.info.size());
    for (uint32_t g = 0; g < n_set; g++) put_u64(mine_tab, kept_index.empty() ? g : kept_index[g]);
    for (const GenomeInfo& gi : sd.info) {
        put_str(mine_tab, gi.file_name); put_u64(mine_tab, gi.contig_order); put_u64(mine_tab, gi.contigs.size());
        put_str(mine_tab, gi.contigs.empty() ? std::string() : gi.contigs[0]);
        uint64_t tot = 0; for (uint32_t l : gi.contig_lengths) tot += l;
        put_u64(mine_tab, tot);
    }
    std::vector<std::string> tabs;
    if (!node_all_gather_v(node, mine_tab, tabs)) die("a rank failed; all ranks stop", true);
    std::vector<GenomeInfo> info_all; std::vector<uint32_t> row_of;                 // row_of[global set genome] = row of info_all, or ~0u
    uint64_t n_info_total = 0;
    for (int r = 0; r < W; r++) {
        Reader rd{tabs[r]};
        const uint64_t c = rd.u64(), k = rd.u64(), m = rd.u64(), ns = rd.u64(), ni = rd.u64();
        if (c != a.c || k != a.k || m != a.m) die("the ranks' sketch parameters differ (sketch files made with different -c / -k / -m?)");
        n_info_total += ni;
        if (rank != 0) continue;
        const uint32_t row0 = (uint32_t)info_all.size();
        for (uint64_t g = 0; g < ns; g++) { const uint64_t x = rd.u64(); row_of.push_back(x == 0xFFFFFFFFull ? ~0u : row0 + (uint32_t)x); }
        for (uint64_t x = 0; x < ni; x++) {
            GenomeInfo gi; gi.file_name = rd.str(); gi.contig_order = rd.u64();
            const uint64_t nc = rd.u64(); const std::string first = rd.str(); const uint64_t tot = rd.u64();
            // (the writers read a genome's first contig name, its number of contigs and its total length: the other names stay on their rank)
            gi.contigs.assign(nc ? nc : 1, std::string()); gi.contigs[0] = first;
            gi.contig_lengths.assign(1, (uint32_t)std::min<uint64_t>(tot, 0xFFFFFFFFull));
            info_all.push_back(std::move(gi));
        }
    }
    if (!n_info_total) die("No genomes/sketches found.");                          // triangle.rs:46-49
    const bool learned = !a.no_learned && a.c >= 70 && !a.individual && !a.median;   // regression.rs:8-10, parse.rs:885-889
    skh_map_params mp = map_params(a, learned);
    const bool rescue_small = !a.faster_small && !a.small_genomes;              // parse.rs:798
    uint32_t *oi = nullptr, *oj = nullptr; skh_ani_result* res = nullptr; uint64_t kept = 0, chained = 0;
    if (const char* kr = getenv("SKH_TUNE_NODE_KILL_RANK")) if (atoi(kr) == rank) raise(SIGKILL);   // (fault injection of the tests: a rank that is gone without a word)
    cx.check(skh_triangle_distributed_ex(cx.c, comm, sd.ss, a.s / 100., rescue_small, &mp, SKH_DIST_ROWS_TO_ROOT, &oi, &oj, &res, &kept, &chained, nullptr), "skh_triangle_distributed");
    g_clock.mark("triangle");
    if (rank == 0) {
        std::vector<PairResult> pr(kept);
        for (uint64_t x = 0; x < kept; x++) pr[x] = PairResult{row_of[oi[x]], row_of[oj[x]], res[x]};   // ref = i, query = j (triangle.rs:98)
        if (a.sparse) emit(a.out, format_sparse(info_all, pr, a.o));
        else {
            std::string ani, af; format_phylip(info_all, pr, a.individual, a.o, ani, af);
            emit(a.out, ani); emit(a.out.empty() ? std::string("skani_matrix.af") : a.out + ".af", af);   // file_io.rs:428,467
        }
        g_clock.mark("write");
    }
    skh_free(oi); skh_free(oj); skh_free(res);
    if (!node_barrier(node)) die("a rank failed; all ranks stop", true);                 // (nobody takes its communicator down while a peer is still inside it)
    skh_comm_destroy(comm);
    skh_sketch_set_destroy(sd.ss);
    g_pinned.release();
    skh_ctx_destroy(cx.c);
    if (rank == 0) g_clock.done();
    fflush(stdout); fflush(stderr);
    _exit(0);                                                                       // (a forked rank: the launcher's static state is not this process's to take down)
}

int run_dist(Args& a, Ctx& cx) {
    std::vector<std::string> q = a.queries, r = a.refs;
    if (!a.qlist.empty()) { auto l = read_list(a.qlist); q.insert(q.end(), l.begin(), l.end()); }
    if (!a.rlist.empty()) { auto l = read_list(a.rlist); r.insert(r.end(), l.begin(), l.end()); }
    if (q.empty() && r.empty() && a.files.size() >= 2) { q.push_back(a.files[0]); r.assign(a.files.begin() + 1, a.files.end()); }   // cli.rs:115-121
    if (q.empty() || r.empty()) die("No reference sketches/genomes or query sketches/genomes found.");
    // sketch-file sides first: their parameters are the ones FASTA sides get sketched with (dist.rs:28-52)
    const bool r_sk = all_sketch_files(r), q_sk = all_sketch_files(q);
    Side lr, lq;
    if (r_sk) lr = load_side(cx, r, a.ri, a);
    const uint32_t rc = a.c, rk = a.k, rm = a.m;
    if (q_sk) lq = load_side(cx, q, a.qi, a);
    if (r_sk && q_sk && (rc != a.c || rk != a.k || rm != a.m)) die("Query sketch parameters were not equal to reference sketch parameters. Exiting.");
    // FASTA sides: the streaming ingest of `triangle` (parser threads -> pinned buffers -> asynchronous copy + pack) when the files allow it.  A set made
    // that way numbers its genomes by the sorted file list, files without a kept contig included: *_row maps a genome to its entry in `info`.
    std::vector<uint32_t> r_row, q_row;
    auto fasta_side = [&](const std::vector<std::string>& files, bool individual, std::vector<uint32_t>& row) {
        Side sd;
        if (!individual) { Streamed st = stream_side(cx, files, a); if (st.ok) { sd.ss = st.ss; sd.info = std::move(st.info); row = std::move(st.kept_index); return sd; } }
        sd = load_side(cx, files, individual, a); g_clock.mark("load_sketch");
        return sd;
    };
    if (!r_sk) lr = fasta_side(r, a.ri, r_row);
    if (!q_sk) lq = fasta_side(q, a.qi, q_row);
    if (lq.info.empty() || lr.info.empty()) die("No reference sketches/genomes or query sketches/genomes found.");
    skh_sketch_set* sq = lq.ss; skh_sketch_set* sr = lr.ss;
    const bool learned = !a.no_learned && a.c >= 70 && !a.qi && !a.ri && !a.median;   // parse.rs:752-756
    skh_map_params mp = map_params(a, learned);
    const bool rescue_small = !a.faster_small && !a.small_genomes;              // parse.rs:636
    const bool index = ((q.size() > 50 || a.qi) && !a.no_marker_index) || a.marker_index;   // parse.rs:750 (FULL_INDEX_THRESH; --marker-index: the pre-0.3 spelling forces it on)
    uint32_t *pq = nullptr, *prf = nullptr; uint64_t np = 0;
    cx.check(skh_screen(cx.c, sr, sq, a.s / 100., index ? SKH_SCREEN_REFS : SKH_SCREEN_QUICK, rescue_small, &pq, &prf, &np), "skh_screen");   // dist.rs:104-122
    std::vector<skh_ani_result> res(np);
    cx.check(skh_chain_pairs(cx.c, sr, sq, prf, pq, np, &mp, res.data(), nullptr), "skh_chain_pairs");   // chain_seeds(ref, query): dist.rs:114
    std::vector<PairResult> pr;
    for (uint64_t x = 0; x < np; x++) if (res[x].ani > 0.1f) pr.push_back(PairResult{r_row.empty() ? prf[x] : r_row[prf[x]], q_row.empty() ? pq[x] : q_row[pq[x]], res[x]});   // dist.rs:115
    skh_free(pq); skh_free(prf);
    g_clock.mark("screen_chain");
    emit(a.out, format_query_ref_list(lr.info, lq.info, pr, a.n, a.o));
    g_clock.mark("write");
    skh_sketch_set_destroy(sq); skh_sketch_set_destroy(sr);
    return 0;
}

int run_sketch(Args& a, Ctx& cx) {                                              // sketch.rs:15-175
    std::vector<std::string> files = a.files;
    if (!a.list.empty()) { auto l = read_list(a.list); files.insert(files.end(), l.begin(), l.end()); }
    if (files.empty()) die("No reference inputs found.");
    if (a.out.empty()) die("sketch needs -o <output folder>");
    if (mkdir(a.out.c_str(), 0777) != 0) die("Output directory exists; output directory must not be an existing directory. Exiting.");   // sketch.rs:19-23
    LoadedGenomes lg = load_genomes(files, a.individual, a.threads);
    if (lg.info.empty()) die("No genomes/sketches found.");
    skh_sketch_set* ss = sketch(cx, lg, a);
    SketchFileParams fp; fp.c = a.c; fp.k = a.k; fp.marker_c = a.m;
    write_sketch_db(a.out, fp, export_blobs(cx, ss, lg.info, a), a.separate_sketches, a.individual);
    skh_sketch_set_destroy(ss);
    return 0;
}

int run_search(Args& a, Ctx& cx) {                                              // search.rs:16-282
    if (a.database.empty()) die("search needs -d <folder written by `sketch`>");
    std::vector<std::string> q = a.queries; q.insert(q.end(), a.files.begin(), a.files.end());
    if (!a.qlist.empty()) { auto l = read_list(a.qlist); q.insert(q.end(), l.begin(), l.end()); }
    if (q.empty()) die("No query sketches/genomes found.");
    SketchDb db;
    try { db = read_sketch_db(a.database); } catch (const std::exception& e) { die(e.what()); }
    if (db.sketches.empty()) die("No reference sketches found in the database folder.");
    a.c = (uint32_t)db.params.c; a.k = (uint32_t)db.params.k; a.m = (uint32_t)db.params.marker_c;   // queries follow the database's parameters (search.rs:37,115-125)
    // the database order is the index order.  The sketches become HBM-resident shards (one sketch-set build handles < 2^32
    // seed positions; --shard-positions bounds a shard), the markers of the whole database one markers-only set that is screened
    // with a single call, and the hits of all shards are chained in one batch.
    std::vector<GenomeInfo> rinfo = infos_of(db.sketches);
    std::vector<skh_sketch_set*> shards; std::vector<uint32_t> shard_of(db.sketches.size()), local_of(db.sketches.size());
    {
        std::vector<SketchBlob> part; uint64_t npos = 0;
        auto flush = [&]() { if (part.empty()) return; shards.push_back(import_blobs(cx, db.params, part, a.seeding_mode)); part.clear(); npos = 0; };
        for (size_t g = 0; g < db.sketches.size(); g++) {
            if (!part.empty() && npos + db.sketches[g].records.size() > a.shard_positions) flush();
            shard_of[g] = (uint32_t)shards.size(); local_of[g] = (uint32_t)part.size();
            npos += db.sketches[g].records.size();
            part.push_back(std::move(db.sketches[g]));
        }
        flush();
    }
    skh_sketch_set* marker_index = shards.size() == 1 ? shards[0] : nullptr;
    if (!marker_index) {
        for (auto& m : db.markers) { m.has_seeds = true; m.records.clear(); m.contig_lengths.assign(1, (uint32_t)std::min<uint64_t>(m.total_sequence_length, 0x7FFF0000ull)); }
        marker_index = import_blobs(cx, db.params, db.markers, a.seeding_mode);   // markers.bin: every genome entered as one contig of its total length
    }
    const uint32_t dc = a.c, dk = a.k, dm = a.m;
    Side lq = load_side(cx, q, a.qi, a);
    if (lq.info.empty()) die("No query sketches/genomes found.");
    if (lq.from_sketch && (dc != a.c || dk != a.k || dm != a.m)) die("Query sketch parameters not equal to reference sketch parameters; no ANI calculated");
    const bool learned = !a.no_learned && a.c >= 70 && !a.qi && !a.median;      // regression.rs:8-10 via search.rs:52
    skh_map_params mp = map_params(a, learned);
    const bool index = (q.size() > 50 || a.qi) && !a.no_marker_index;           // parse.rs:960
    uint32_t *pq = nullptr, *prf = nullptr; uint64_t np = 0;
    // search.rs:126-147: check_markers_quickly(query, ref, s, false) or screen_refs_indices
    cx.check(skh_screen(cx.c, marker_index, lq.ss, a.s / 100., index ? SKH_SCREEN_REFS_INDICES : SKH_SCREEN_QUICK, 0, &pq, &prf, &np), "skh_screen");
    std::vector<skh_ani_result> res(np);
    std::vector<uint32_t> pset(np), pref(np);
    for (uint64_t x = 0; x < np; x++) { pset[x] = shard_of[prf[x]]; pref[x] = local_of[prf[x]]; }
    cx.check(skh_chain_pairs_multi(cx.c, shards.data(), (uint32_t)shards.size(), lq.ss, pset.data(), pref.data(), pq, np, &mp, res.data()), "skh_chain_pairs_multi");
    std::vector<PairResult> pr;
    for (uint64_t x = 0; x < np; x++) if (res[x].ani > 0.5f) pr.push_back(PairResult{prf[x], pq[x], res[x]});   // search.rs:178
    skh_free(pq); skh_free(prf);
    emit(a.out, format_query_ref_list(rinfo, lq.info, pr, a.n, a.o));
    skh_sketch_set_destroy(lq.ss);
    if (shards.size() > 1) skh_sketch_set_destroy(marker_index);
    for (auto* sh : shards) skh_sketch_set_destroy(sh);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    Args a = parse(argc, argv);
    if (a.gpus < 1 || a.gpus > 64) die("--gpus: between 1 and 64");
    if (a.gpus > 1 || (a.cmd == "triangle" && getenv("SKANI_HIP_FORCE_NODE"))) {     // (SKANI_HIP_FORCE_NODE: the rank driver with a world of one -- a forked rank, the RCCL communicator and its self-test on a one-GPU box)
        if (a.cmd != "triangle") die("--gpus is for `triangle` (dist and search run on one GPU)");
        try { return run_triangle_node(a, argv[0]); } catch (const std::exception& e) { die(e.what()); }
    }
    Ctx cx;
    if (skh_ctx_create(a.device, &cx.c) != 0) die("no usable MI355X device (skani-hip has no CPU path)");
    const std::string md = models_dir(a, argv[0]);
    cx.check(skh_load_models(cx.c, (md + "/gbdt_c125.bin").c_str(), (md + "/gbdt_c200.bin").c_str()), "skh_load_models");
    g_clock.mark("startup");
    int rc;
    if (a.cmd == "triangle") rc = run_triangle(a, cx);
    else if (a.cmd == "dist") rc = run_dist(a, cx);
    else if (a.cmd == "sketch") rc = run_sketch(a, cx);
    else if (a.cmd == "search") rc = run_search(a, cx);
    else die("unknown subcommand " + a.cmd + " (supported: triangle, dist, sketch, search)");
    g_pinned.release();
    skh_ctx_destroy(cx.c);
    g_clock.done();
    return rc;
}
