// capi_test.cpp -- TEST-ONLY: extern "C" hooks so the CPU test-suite can exercise the C++ host code (FASTA reader, writers, formats) via ctypes.
// Built by tests/host_shims/build_shims.py into tests/host_shims/libskani_host_test.so together with the host sources; not part of the product libraries.
#include <cstdlib>
#include <cstring>

#include "../../skani_amd/host/host.hpp"

using namespace skhost;

static char* dup(const std::string& s) { char* p = (char*)malloc(s.size() + 1); memcpy(p, s.data(), s.size()); p[s.size()] = 0; return p; }

static std::vector<GenomeInfo> infos(uint32_t n, const char** files, const char** ctg0, const uint32_t* total_len) {
    std::vector<GenomeInfo> g(n);
    for (uint32_t i = 0; i < n; i++) { g[i].file_name = files[i]; g[i].contigs.push_back(ctg0[i]); g[i].contig_lengths.push_back(total_len[i]); }
    return g;
}
static OutOpts opts(uint32_t flags) { OutOpts o; o.ci = flags & 1; o.detailed = flags & 2; o.short_header = flags & 4; o.diagonal = flags & 8; o.full_matrix = flags & 16; o.distance = flags & 32; return o; }

extern "C" {
void skhost_test_free(char* p) { free(p); }

// returns "name\tlen\n" per kept record (>= min_len) -- checks reader semantics
char* skhost_fasta_summary(const char* path, uint64_t min_len) {
    try {
        std::string s;
        for (auto& r : read_fasta(path)) if (r.seq.size() >= min_len) s += r.name + "\t" + std::to_string(r.seq.size()) + "\n";
        return dup(s);
    } catch (const std::exception& e) { return dup(std::string("ERROR ") + e.what()); }
}
// the mapped-file parser of the streaming ingest: "name\tlen\n" per kept contig, then "=" and the kept bases as they were written; "NOT PLAIN" for gzip / FASTQ
char* skhost_fasta_plain_threads(const char* path, uint64_t min_len, int threads);
char* skhost_fasta_plain(const char* path, uint64_t min_len) { return skhost_fasta_plain_threads(path, min_len, 1); }
// threads > 1: the several-thread parse whatever the file's size
char* skhost_fasta_plain_threads(const char* path, uint64_t min_len, int threads) {
    try {
        FILE* f = fopen(path, "rb"); if (!f) return dup("ERROR cannot open");
        fseek(f, 0, SEEK_END); const long n = ftell(f); fclose(f);
        std::vector<uint8_t> dst((size_t)n + 16); size_t used = 0; std::vector<std::string> names; std::vector<uint64_t> lens;
        if (!parse_fasta_plain(path, dst.data(), dst.size(), (size_t)min_len, &used, names, lens, threads, 0)) return dup("NOT PLAIN");
        std::string s;
        for (size_t i = 0; i < names.size(); i++) s += names[i] + "\t" + std::to_string(lens[i]) + "\n";
        return dup(s + "=" + std::string((const char*)dst.data(), used));
    } catch (const std::exception& e) { return dup(std::string("ERROR ") + e.what()); }
}
char* skhost_fasta_seq(const char* path, uint32_t idx) {
    try { auto v = read_fasta(path); return dup(idx < v.size() ? v[idx].seq : std::string()); } catch (const std::exception& e) { return dup(std::string("ERROR ") + e.what()); }
}

char* skhost_phylip(uint32_t n, const char** files, const char** ctg0, const uint32_t* total_len, uint32_t n_res, const uint32_t* ri, const uint32_t* qi,
                    const skh_ani_result* res, uint32_t flags, int use_contig_names, int want_af) {
    auto g = infos(n, files, ctg0, total_len);
    std::vector<PairResult> pr(n_res); for (uint32_t x = 0; x < n_res; x++) pr[x] = PairResult{ri[x], qi[x], res[x]};
    std::string a, f; format_phylip(g, pr, use_contig_names != 0, opts(flags), a, f);
    return dup(want_af ? f : a);
}
char* skhost_sparse(uint32_t n, const char** files, const char** ctg0, const uint32_t* total_len, uint32_t n_res, const uint32_t* ri, const uint32_t* qi,
                    const skh_ani_result* res, uint32_t flags) {
    auto g = infos(n, files, ctg0, total_len);
    std::vector<PairResult> pr(n_res); for (uint32_t x = 0; x < n_res; x++) pr[x] = PairResult{ri[x], qi[x], res[x]};
    return dup(format_sparse(g, pr, opts(flags)));
}
char* skhost_query_ref_list(uint32_t nr, const char** rfiles, const char** rctg0, const uint32_t* rlen, uint32_t nq, const char** qfiles, const char** qctg0,
                            const uint32_t* qlen, uint32_t n_res, const uint32_t* ri, const uint32_t* qi, const skh_ani_result* res, uint64_t n_max, uint32_t flags) {
    auto r = infos(nr, rfiles, rctg0, rlen); auto q = infos(nq, qfiles, qctg0, qlen);
    std::vector<PairResult> pr(n_res); for (uint32_t x = 0; x < n_res; x++) pr[x] = PairResult{ri[x], qi[x], res[x]};
    return dup(format_query_ref_list(r, q, pr, (size_t)n_max, opts(flags)));
}

// ---- on-disk formats (formats.cpp) ----
static uint64_t blob_checksum(const SketchBlob& b) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    for (auto& r : b.records) { mix(r.seed); mix(r.pos); mix(r.ctgcanon); }
    for (auto m : b.markers) mix(m);
    for (auto l : b.contig_lengths) mix(l);
    mix(b.total_sequence_length);
    return h;
}
static std::string blob_line(const SketchBlob& b) {
    return b.file_name + "\t" + std::to_string(b.records.size()) + "\t" + std::to_string(b.markers.size()) + "\t" + std::to_string(b.contigs.size()) + "\t" +
           std::to_string(b.total_sequence_length) + "\t" + std::to_string(b.contig_order) + "\t" + std::to_string(blob_checksum(b)) + "\n";
}

// decode one .sketch file; arrays are malloc'd (skhost_free them).  Returns NULL on success, else an error string.
char* skhost_sketch_read(const char* path, uint64_t* ckm /*c,k,marker_c*/, int* format, uint64_t* n_rec, uint32_t** seed, uint32_t** pos, uint32_t** cc,
                         uint64_t* n_markers, uint64_t** markers, uint64_t* n_contigs, uint32_t** contig_lengths, uint64_t* scalars /*total_len, marker_c, c, k, contig_order, repetitive*/,
                         char** names /* file_name, then contig names, '\n'-separated */) {
    try {
        FILE* f = fopen(path, "rb"); if (!f) return dup("cannot open file");
        std::string bytes; char buf[1 << 16]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) bytes.append(buf, k); fclose(f);
        SketchFileParams sp; SketchBlob b;
        decode_sketch((const uint8_t*)bytes.data(), bytes.size(), sp, b, format);
        ckm[0] = sp.c; ckm[1] = sp.k; ckm[2] = sp.marker_c;
        *n_rec = b.records.size(); *n_markers = b.markers.size(); *n_contigs = b.contig_lengths.size();
        *seed = (uint32_t*)malloc(4 * b.records.size() + 4); *pos = (uint32_t*)malloc(4 * b.records.size() + 4); *cc = (uint32_t*)malloc(4 * b.records.size() + 4);
        for (size_t i = 0; i < b.records.size(); i++) { (*seed)[i] = b.records[i].seed; (*pos)[i] = b.records[i].pos; (*cc)[i] = b.records[i].ctgcanon; }
        *markers = (uint64_t*)malloc(8 * b.markers.size() + 8); memcpy(*markers, b.markers.data(), 8 * b.markers.size());
        *contig_lengths = (uint32_t*)malloc(4 * b.contig_lengths.size() + 4); memcpy(*contig_lengths, b.contig_lengths.data(), 4 * b.contig_lengths.size());
        scalars[0] = b.total_sequence_length; scalars[1] = b.marker_c; scalars[2] = b.c; scalars[3] = b.k; scalars[4] = b.contig_order; scalars[5] = b.repetitive_kmers;
        std::string nm = b.file_name; for (auto& c : b.contigs) nm += "\n" + c;
        *names = dup(nm);
        return nullptr;
    } catch (const std::exception& e) { return dup(e.what()); }
}

char* skhost_sketch_write(const char* path, const uint64_t* ckm, const char* file_name, uint64_t n_rec, const uint32_t* seed, const uint32_t* pos, const uint32_t* cc,
                          uint64_t n_markers, const uint64_t* markers, uint64_t n_contigs, const char** contig_names, const uint32_t* contig_lengths, uint64_t total_len,
                          uint64_t contig_order) {
    try {
        SketchFileParams sp; sp.c = ckm[0]; sp.k = ckm[1]; sp.marker_c = ckm[2];
        SketchBlob b; b.file_name = file_name; b.records.resize(n_rec);
        for (uint64_t i = 0; i < n_rec; i++) b.records[i] = SeedRecord{seed[i], pos[i], cc[i]};
        b.markers.assign(markers, markers + n_markers); b.contig_lengths.assign(contig_lengths, contig_lengths + n_contigs);
        for (uint64_t i = 0; i < n_contigs; i++) b.contigs.push_back(contig_names[i]);
        b.total_sequence_length = total_len; b.marker_c = sp.c; b.c = sp.c; b.k = sp.k; b.contig_order = contig_order;
        const std::string bytes = encode_sketch(sp, b);
        FILE* f = fopen(path, "wb"); if (!f) return dup("cannot write file");
        fwrite(bytes.data(), 1, bytes.size(), f); fclose(f);
        return nullptr;
    } catch (const std::exception& e) { return dup(e.what()); }
}

// .sketch files -> database folder (which must exist); then skhost_db_summary lists what a reader gets back
char* skhost_db_write(const char* dir, uint32_t n, const char** sketch_files, int separate_files) {
    try {
        SketchFileParams sp; std::vector<std::string> files(sketch_files, sketch_files + n);
        auto blobs = read_sketch_files(files, sp);
        write_sketch_db(dir, sp, blobs, separate_files != 0, false);
        return nullptr;
    } catch (const std::exception& e) { return dup(e.what()); }
}
char* skhost_db_summary(const char* dir) {
    try {
        SketchDb db = read_sketch_db(dir);
        std::string s = "params\t" + std::to_string(db.params.c) + "\t" + std::to_string(db.params.k) + "\t" + std::to_string(db.params.marker_c) + "\n";
        for (auto& b : db.sketches) s += blob_line(b);
        s += "markers\n";
        for (auto& b : db.markers) s += blob_line(b);
        return dup(s);
    } catch (const std::exception& e) { return dup(std::string("ERROR ") + e.what()); }
}
char* skhost_sketch_summary(const char* path) {
    try {
        SketchFileParams sp; auto v = read_sketch_files({path}, sp);
        return dup(v.empty() ? std::string("ERROR unreadable") : blob_line(v[0]));
    } catch (const std::exception& e) { return dup(std::string("ERROR ") + e.what()); }
}
}
