// capi_db.cpp -- extern "C" access to skani's on-disk formats (formats.cpp) for bindings that sit above the C ABI
// (skani_amd/formats.py): a database folder or a list of .sketch files comes back as the flat arrays skh_sketch_import takes.
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "host.hpp"

using namespace skhost;

namespace {

struct FlatDb {
    SketchFileParams sp;
    std::vector<SketchBlob> sketches;
    uint64_t P = 0, M = 0, NC = 0; std::string names;     // names: per sketch file_name '\n' then one line per contig
};

char* dup_str(const std::string& s) { char* p = (char*)malloc(s.size() + 1); memcpy(p, s.data(), s.size()); p[s.size()] = 0; return p; }

void finish(FlatDb* db) {
    for (const SketchBlob& b : db->sketches) {
        db->P += b.records.size(); db->M += b.markers.size(); db->NC += b.contig_lengths.size();
        db->names += b.file_name; db->names.push_back('\n');
        for (size_t c = 0; c < b.contig_lengths.size(); c++) { db->names += c < b.contigs.size() ? b.contigs[c] : std::string(); db->names.push_back('\n'); }
    }
}

}  // namespace

extern "C" {

void skhost_free(char* p) { free(p); }

// kind 0: folder written by `skani sketch` / `skani-hip sketch` (or its markers.bin); kind 1: '\n'-separated .sketch files.
// Returns NULL and sets *out, or an error string (skhost_free it).
char* skhost_db_open(const char* path, int kind, void** out) {
    *out = nullptr;
    try {
        FlatDb* db = new FlatDb();
        if (kind == 0) { SketchDb d = read_sketch_db(path); db->sp = d.params; db->sketches = std::move(d.sketches); }
        else {
            std::vector<std::string> files; std::string cur;
            for (const char* p = path;; p++) { if (*p == '\n' || *p == 0) { if (!cur.empty()) files.push_back(cur); cur.clear(); if (*p == 0) break; } else cur.push_back(*p); }
            db->sketches = read_sketch_files(files, db->sp);
        }
        for (auto& b : db->sketches) if (!b.has_seeds) { delete db; return dup_str("markers-only sketch cannot be loaded for alignment"); }
        finish(db);
        *out = db;
        return nullptr;
    } catch (const std::exception& e) { return dup_str(e.what()); }
}
// dims: n sketches, seed positions, markers, contigs, c, k, marker_c, bytes of the names blob
void skhost_db_dims(const void* h, uint64_t* dims) {
    const FlatDb* db = (const FlatDb*)h;
    dims[0] = db->sketches.size(); dims[1] = db->P; dims[2] = db->M; dims[3] = db->NC; dims[4] = db->sp.c; dims[5] = db->sp.k; dims[6] = db->sp.marker_c;
    dims[7] = db->names.size();
}
void skhost_db_fill(const void* h, uint64_t* pos_off, uint32_t* seed, uint32_t* pos, uint32_t* cc, uint64_t* mk_off, uint64_t* markers, uint64_t* ctg_off,
                    uint32_t* contig_lengths, uint64_t* total_len, uint64_t* contig_order, char* names) {
    const FlatDb* db = (const FlatDb*)h;
    uint64_t p = 0, m = 0, c = 0; size_t g = 0;
    pos_off[0] = mk_off[0] = ctg_off[0] = 0;
    for (const SketchBlob& b : db->sketches) {
        for (const SeedRecord& r : b.records) { seed[p] = r.seed; pos[p] = r.pos; cc[p] = r.ctgcanon; p++; }
        if (!b.markers.empty()) memcpy(markers + m, b.markers.data(), b.markers.size() * 8);
        m += b.markers.size();
        if (!b.contig_lengths.empty()) memcpy(contig_lengths + c, b.contig_lengths.data(), b.contig_lengths.size() * 4);
        c += b.contig_lengths.size();
        total_len[g] = b.total_sequence_length; contig_order[g] = b.contig_order; g++;
        pos_off[g] = p; mk_off[g] = m; ctg_off[g] = c;
    }
    memcpy(names, db->names.data(), db->names.size());
}
void skhost_db_close(void* h) { delete (FlatDb*)h; }

// the inverse: flat arrays (as exported by skh_sketch_export_flat) -> folder.  names as in skhost_db_fill.
char* skhost_db_save(const char* dir, int separate_files, int individual_contig, const uint64_t* ckm, uint64_t n, const uint64_t* pos_off, const uint32_t* seed,
                     const uint32_t* pos, const uint32_t* cc, const uint64_t* mk_off, const uint64_t* markers, const uint64_t* ctg_off,
                     const uint32_t* contig_lengths, const uint64_t* total_len, const uint64_t* contig_order, const char* names) {
    try {
        SketchFileParams sp; sp.c = ckm[0]; sp.k = ckm[1]; sp.marker_c = ckm[2];
        std::vector<SketchBlob> blobs(n);
        const char* at = names;
        auto line = [&]() { const char* e = strchr(at, '\n'); if (!e) throw std::runtime_error("names blob too short"); std::string s(at, e); at = e + 1; return s; };
        for (uint64_t g = 0; g < n; g++) {
            SketchBlob& b = blobs[g];
            b.file_name = line();
            for (uint64_t c = ctg_off[g]; c < ctg_off[g + 1]; c++) { b.contigs.push_back(line()); b.contig_lengths.push_back(contig_lengths[c]); }
            b.records.resize(pos_off[g + 1] - pos_off[g]);
            for (uint64_t i = pos_off[g]; i < pos_off[g + 1]; i++) b.records[i - pos_off[g]] = SeedRecord{seed[i], pos[i], cc[i]};
            b.markers.assign(markers + mk_off[g], markers + mk_off[g + 1]);
            b.total_sequence_length = total_len[g]; b.contig_order = contig_order ? contig_order[g] : 0;
            b.marker_c = sp.c; b.c = sp.c; b.k = sp.k;                                      // Sketch::new stores c in marker_c (types.rs:346)
        }
        write_sketch_db(dir, sp, blobs, separate_files != 0, individual_contig != 0);
        return nullptr;
    } catch (const std::exception& e) { return dup_str(e.what()); }
}

}
