// formats.cpp -- skani's on-disk sketch formats, so that databases written by `skani sketch` load into HBM
// and databases written here are readable by skani.
//
// Everything is bincode 1.3.3 with default options (file_io.rs:696,722; sketch.rs:85,101; sketch_db.rs:47,76,91,117):
// little-endian fixed-width integers, usize -> u64, String / Vec / map / set = u64 length + items, Option = u8 tag, bool = u8,
// tuples and structs = fields in declaration order.
//
//   <name>.sketch      = (SketchParams, Sketch)                       sketch.rs:60-85
//   markers.bin        = (SketchParams, Vec<Sketch markers-only>)     sketch.rs:90-101, types.rs:322-340
//   sketches.db        = concatenated (SketchParams, Sketch) blobs    sketch_db.rs:45-66
//   index.db           = Vec<IndexEntry{file_name, offset, length}>   sketch_db.rs:10-15,68-81
//
// SketchParams (params.rs:136-146): c, k, marker_c: usize; use_syncs, use_aa: bool; acgt_to_aa_encoding: Vec<u64>;
//   acgt_to_aa_letters: Vec<u8>; orf_size: usize.
// Sketch, v0.3 (types.rs:252-277): file_name; kmer_seeds_k: Option<HashMap<u32, u64 tagged index>>; multi_position_storage:
//   Vec<SmallVec<[SeedPosition{pos: u32, contig_index_canonical: u32}; 3]>>; contigs: Vec<String>; total_sequence_length;
//   contig_lengths: Vec<u32>; repetitive_kmers; marker_seeds: HashSet<u64>; marker_c; c; k; contig_order;
//   individual_contig: bool; amino_acid: bool.
//   Tagged index (types.rs:201-243): bit 0 set = one position packed as (pos << 31 | contig_index_canonical) << 1 | 1;
//   bit 0 clear = (index into multi_position_storage) << 1.
// Sketch, pre-0.3 (the layout of the reference's bundled test_files/e.coli-o157.fasta.sketch; skani >= 0.3 refuses it,
//   file_io.rs:703-708 -- read here because it is the reference's golden seed set): file_name; Option<HashMap<u32,
//   Vec<{pos: u32, canonical: bool, contig_index: u32, phase: u8}>>>; contigs; total_sequence_length; contig_lengths;
//   repetitive_kmers; marker_seeds; marker_c; c; k; contig_order; amino_acid.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <unordered_map>

#include "host.hpp"

namespace skhost {

namespace {

struct Reader {
    const uint8_t* p; size_t n, at = 0;
    Reader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    void need(size_t k) const { if (k > n - at) throw std::runtime_error("truncated or corrupt bincode stream"); }
    uint8_t u8() { need(1); return p[at++]; }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, p + at, 4); at += 4; return v; }
    uint64_t u64() { need(8); uint64_t v; memcpy(&v, p + at, 8); at += 8; return v; }
    bool boolean() { uint8_t b = u8(); if (b > 1) throw std::runtime_error("corrupt bincode stream (bool)"); return b != 0; }
    // a length that is followed by at least `item` bytes per element: rejects absurd lengths before any allocation
    uint64_t len(size_t item) { uint64_t l = u64(); if (item && l > (n - at) / item) throw std::runtime_error("corrupt bincode stream (length)"); return l; }
    std::string str() { uint64_t l = len(1); std::string s((const char*)p + at, (size_t)l); at += l; return s; }
};

struct Writer {
    std::string out;
    void u8(uint8_t v) { out.push_back((char)v); }
    void u32(uint32_t v) { out.append((const char*)&v, 4); }
    void u64(uint64_t v) { out.append((const char*)&v, 8); }
    void str(const std::string& s) { u64(s.size()); out += s; }
};

// The standard genetic code indexed by a codon in A,C,G,T base-4 order (types.rs:28-29 holds the same table) and the
// amino-acid numbering of params.rs:151-174 ('R' is listed twice there; the later entry, 15, wins in the collected map).
void aa_tables(std::vector<uint64_t>& enc, std::vector<uint8_t>& letters) {
    static const char* tcag = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";   // NCBI table 1, T,C,A,G order
    static const int to_tcag[4] = {2, 1, 3, 0};                                                      // A,C,G,T -> index in T,C,A,G
    static const char* order = "ARNDCEFGHIKLMPQRSTVWY";
    letters.assign(64, 0); enc.assign(64, 0);
    for (int i = 0; i < 64; i++) {
        const int a = i >> 4, b = (i >> 2) & 3, c = i & 3;
        const char aa = tcag[to_tcag[a] * 16 + to_tcag[b] * 4 + to_tcag[c]];
        letters[i] = (uint8_t)aa;
        uint64_t code = 21;                                                                          // STOP_CODON, params.rs:14
        for (int j = 0; order[j]; j++) if (order[j] == aa) code = (uint64_t)j;                       // last match wins => R = 15
        enc[i] = code;
    }
}

void write_params(Writer& w, const SketchFileParams& sp) {
    std::vector<uint64_t> enc; std::vector<uint8_t> letters; aa_tables(enc, letters);
    w.u64(sp.c); w.u64(sp.k); w.u64(sp.marker_c); w.u8(sp.use_syncs); w.u8(sp.use_aa);
    w.u64(64); for (uint64_t v : enc) w.u64(v);
    w.u64(64); for (uint8_t v : letters) w.u8(v);
    w.u64(30);                                                                                       // ORF_SIZE, params.rs:32
}

void read_params(Reader& r, SketchFileParams& sp) {
    sp.c = r.u64(); sp.k = r.u64(); sp.marker_c = r.u64(); sp.use_syncs = r.boolean(); sp.use_aa = r.boolean();
    uint64_t n = r.len(8); r.need(n * 8); r.at += n * 8;
    n = r.len(1); r.need(n); r.at += n;
    r.u64();
    if (sp.c == 0 || sp.k == 0 || sp.k > 16 || sp.marker_c < sp.c) throw std::runtime_error("implausible sketch parameters (not a skani sketch?)");
}

void sort_records(std::vector<SeedRecord>& v) {
    std::sort(v.begin(), v.end(), [](const SeedRecord& a, const SeedRecord& b) {
        const uint32_t ca = a.ctgcanon >> 1, cb = b.ctgcanon >> 1;
        return ca != cb ? ca < cb : a.pos < b.pos;
    });
}

void write_sketch_body(Writer& w, const SketchBlob& s) {
    w.str(s.file_name);
    if (!s.has_seeds) { w.u8(0); w.u64(0); }
    else {
        // replay add_seed_position (types.rs:281-305) over the records in (contig, pos) order: a seed's second position opens
        // a multi_position_storage entry, later ones append to it
        std::vector<std::vector<SeedRecord>> multi;
        std::vector<std::pair<uint32_t, uint64_t>> entries;                       // (seed, tagged) in first-seen order
        std::unordered_map<uint32_t, size_t> where; where.reserve(s.records.size() * 2);
        for (const SeedRecord& rec : s.records) {
            auto it = where.find(rec.seed);
            if (it == where.end()) {
                where.emplace(rec.seed, entries.size());
                entries.emplace_back(rec.seed, ((((uint64_t)rec.pos << 31) | rec.ctgcanon) << 1) | 1ull);
            } else {
                uint64_t& tagged = entries[it->second].second;
                if (tagged & 1ull) {
                    const uint64_t packed = tagged >> 1;
                    SeedRecord first{rec.seed, (uint32_t)(packed >> 31), (uint32_t)(packed & 0x7FFFFFFFull)};
                    tagged = (uint64_t)multi.size() << 1;
                    multi.push_back({first, rec});
                } else multi[(size_t)(tagged >> 1)].push_back(rec);
            }
        }
        w.u8(1); w.u64(entries.size());
        for (auto& e : entries) { w.u32(e.first); w.u64(e.second); }
        w.u64(multi.size());
        for (auto& m : multi) { w.u64(m.size()); for (auto& rec : m) { w.u32(rec.pos); w.u32(rec.ctgcanon); } }
    }
    w.u64(s.contigs.size()); for (auto& c : s.contigs) w.str(c);
    w.u64(s.total_sequence_length);
    w.u64(s.contig_lengths.size()); for (uint32_t l : s.contig_lengths) w.u32(l);
    w.u64(s.repetitive_kmers);
    w.u64(s.markers.size()); for (uint64_t m : s.markers) w.u64(m);
    w.u64(s.marker_c); w.u64(s.c); w.u64(s.k); w.u64(s.contig_order);
    w.u8(s.individual_contig); w.u8(s.amino_acid);
}

void read_tail(Reader& r, SketchBlob& s, bool legacy) {
    uint64_t n = r.len(8); s.contigs.resize(n); for (auto& c : s.contigs) c = r.str();
    s.total_sequence_length = r.u64();
    n = r.len(4); s.contig_lengths.resize(n); for (auto& l : s.contig_lengths) l = r.u32();
    s.repetitive_kmers = r.u64();
    n = r.len(8); s.markers.resize(n); for (auto& m : s.markers) m = r.u64();
    std::sort(s.markers.begin(), s.markers.end());
    s.marker_c = r.u64(); s.c = r.u64(); s.k = r.u64(); s.contig_order = r.u64();
    s.individual_contig = legacy ? false : r.boolean();
    s.amino_acid = r.boolean();
}

void read_sketch_body(Reader& r, SketchBlob& s) {
    s.file_name = r.str();
    s.records.clear();
    const uint8_t tag = r.u8();
    if (tag > 1) throw std::runtime_error("corrupt bincode stream (option tag)");
    s.has_seeds = tag == 1;
    std::vector<std::pair<uint32_t, uint64_t>> entries;
    if (tag) { uint64_t n = r.len(12); entries.resize(n); for (auto& e : entries) { e.first = r.u32(); e.second = r.u64(); } }
    uint64_t nm = r.len(8);
    std::vector<std::pair<size_t, uint64_t>> multi(nm);                            // (byte offset of the items, count)
    for (auto& m : multi) { uint64_t l = r.len(8); m = {r.at, l}; r.at += l * 8; }
    s.records.reserve(entries.size() + entries.size() / 8);
    for (auto& e : entries) {
        if (e.second & 1ull) {
            const uint64_t packed = e.second >> 1;
            s.records.push_back(SeedRecord{e.first, (uint32_t)(packed >> 31), (uint32_t)(packed & 0x7FFFFFFFull)});
        } else {
            const uint64_t i = e.second >> 1;
            if (i >= nm) throw std::runtime_error("corrupt sketch (multi-position index out of range)");
            const uint8_t* q = r.p + multi[i].first;
            for (uint64_t x = 0; x < multi[i].second; x++) { SeedRecord rec; rec.seed = e.first; memcpy(&rec.pos, q + x * 8, 4); memcpy(&rec.ctgcanon, q + x * 8 + 4, 4); s.records.push_back(rec); }
        }
    }
    sort_records(s.records);
    read_tail(r, s, false);
}

void read_sketch_body_legacy(Reader& r, SketchBlob& s) {
    s.file_name = r.str();
    s.records.clear();
    const uint8_t tag = r.u8();
    if (tag > 1) throw std::runtime_error("corrupt bincode stream (option tag)");
    s.has_seeds = tag == 1;
    if (tag) {
        uint64_t n = r.len(12);
        s.records.reserve(n + n / 8);
        for (uint64_t i = 0; i < n; i++) {
            const uint32_t seed = r.u32(); const uint64_t l = r.len(10);
            for (uint64_t x = 0; x < l; x++) {
                const uint32_t pos = r.u32(); const bool canon = r.boolean(); const uint32_t ctg = r.u32(); r.u8();
                if (ctg >= (1u << 30)) throw std::runtime_error("corrupt sketch (contig index)");
                s.records.push_back(SeedRecord{seed, pos, (ctg << 1) | (canon ? 1u : 0u)});
            }
        }
    }
    sort_records(s.records);
    read_tail(r, s, true);
}

void validate(const SketchBlob& s) {
    if (!s.has_seeds) return;
    if (s.contig_lengths.size() != s.contigs.size()) throw std::runtime_error("corrupt sketch " + s.file_name + " (contig tables differ in length)");
    for (const SeedRecord& rec : s.records) {
        const uint32_t ctg = rec.ctgcanon >> 1;
        if (ctg >= s.contig_lengths.size() || rec.pos >= s.contig_lengths[ctg]) throw std::runtime_error("corrupt sketch " + s.file_name + " (seed position outside its contig)");
    }
}

std::string slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string s; char buf[1 << 16]; size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, k);
    fclose(f);
    return s;
}

void spill(const std::string& path, const std::string& bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    const size_t k = fwrite(bytes.data(), 1, bytes.size(), f);
    if (fclose(f) != 0 || k != bytes.size()) throw std::runtime_error("short write to " + path);
}

bool exists(const std::string& path) { FILE* f = fopen(path.c_str(), "rb"); if (f) fclose(f); return f != nullptr; }

std::string base_name(const std::string& p) { size_t s = p.find_last_of('/'); return s == std::string::npos ? p : p.substr(s + 1); }

}  // namespace

std::string encode_sketch(const SketchFileParams& sp, const SketchBlob& s) { Writer w; write_params(w, sp); write_sketch_body(w, s); return std::move(w.out); }

size_t decode_sketch(const uint8_t* bytes, size_t n, SketchFileParams& sp, SketchBlob& s, int* format) {
    // v0.3 first; the pre-0.3 layout only if that fails (the two differ from the first map value on)
    try {
        Reader r(bytes, n); read_params(r, sp); read_sketch_body(r, s); validate(s);
        if (r.at != n) throw std::runtime_error("trailing bytes after the sketch");
        if (format) *format = 3;
        return r.at;
    } catch (const std::runtime_error& first) {
        try {
            Reader r(bytes, n); read_params(r, sp); read_sketch_body_legacy(r, s); validate(s);
            if (r.at != n) throw std::runtime_error("trailing bytes after the sketch");
            if (format) *format = 2;
            return r.at;
        } catch (const std::runtime_error&) { throw first; }
    }
}

std::string encode_markers(const SketchFileParams& sp, const std::vector<SketchBlob>& sketches) {
    Writer w; write_params(w, sp); w.u64(sketches.size());
    for (const SketchBlob& s : sketches) {                                          // Sketch::get_markers_only, types.rs:322-340
        SketchBlob m = s; m.has_seeds = false; m.records.clear(); m.contig_lengths.clear();
        write_sketch_body(w, m);
    }
    return std::move(w.out);
}

void decode_markers(const uint8_t* bytes, size_t n, SketchFileParams& sp, std::vector<SketchBlob>& out) {
    Reader r(bytes, n); read_params(r, sp);
    const uint64_t cnt = r.len(8 * 12);
    out.resize(cnt);
    for (auto& s : out) read_sketch_body(r, s);
}

std::string encode_index(const std::vector<IndexEntry>& idx) {
    Writer w; w.u64(idx.size());
    for (auto& e : idx) { w.str(e.file_name); w.u64(e.offset); w.u64(e.length); }
    return std::move(w.out);
}

std::vector<IndexEntry> decode_index(const uint8_t* bytes, size_t n) {
    Reader r(bytes, n); const uint64_t cnt = r.len(24);
    std::vector<IndexEntry> v(cnt);
    for (auto& e : v) { e.file_name = r.str(); e.offset = r.u64(); e.length = r.u64(); }
    return v;
}

void write_sketch_db(const std::string& dir, const SketchFileParams& sp, const std::vector<SketchBlob>& sketches, bool separate_files, bool individual_contig) {
    if (separate_files) {                                                            // sketch.rs:60-85
        for (size_t j = 0; j < sketches.size(); j++) {
            const SketchBlob& s = sketches[j];
            const std::string fn = base_name(s.file_name);
            spill(dir + "/" + (individual_contig ? std::to_string(s.contig_order) + "_" + fn : fn) + ".sketch", encode_sketch(sp, s));
        }
    } else {                                                                         // sketch_db.rs:45-81
        FILE* f = fopen((dir + "/sketches.db").c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + dir + "/sketches.db");
        std::vector<IndexEntry> idx; uint64_t off = 0;
        for (const SketchBlob& s : sketches) {
            const std::string b = encode_sketch(sp, s);
            if (fwrite(b.data(), 1, b.size(), f) != b.size()) { fclose(f); throw std::runtime_error("short write to sketches.db"); }
            idx.push_back(IndexEntry{s.file_name, off, (uint64_t)b.size()}); off += b.size();
        }
        if (fclose(f) != 0) throw std::runtime_error("short write to sketches.db");
        spill(dir + "/index.db", encode_index(idx));
    }
    spill(dir + "/markers.bin", encode_markers(sp, sketches));
}

std::vector<SketchBlob> read_sketch_files(const std::vector<std::string>& files, SketchFileParams& sp) {
    std::vector<SketchBlob> out;
    for (const std::string& f : files) {                                             // file_io.rs:680-729
        if (f.find("markers.bin") != std::string::npos) continue;
        const std::string bytes = slurp(f);
        SketchBlob s; SketchFileParams p;
        try { decode_sketch((const uint8_t*)bytes.data(), bytes.size(), p, s, nullptr); }
        catch (const std::runtime_error& e) { fprintf(stderr, "ERROR %s is not a valid .sketch file or is corrupted (%s).\n", f.c_str(), e.what()); continue; }
        sp = p; out.push_back(std::move(s));
    }
    std::stable_sort(out.begin(), out.end(), [](const SketchBlob& a, const SketchBlob& b) { return a.file_name < b.file_name; });
    return out;
}

SketchDb read_sketch_db(const std::string& dir_or_marker_file) {
    std::string dir = dir_or_marker_file;
    if (dir.size() >= 11 && dir.compare(dir.size() - 11, 11, "markers.bin") == 0) dir = dir.size() > 12 ? dir.substr(0, dir.size() - 12) : std::string(".");
    while (dir.size() > 1 && dir.back() == '/') dir.pop_back();
    SketchDb db;
    if (!exists(dir + "/markers.bin")) throw std::runtime_error("markers.bin not found in the folder. Ensure that the folder was generated by `skani sketch`.");   // search.rs:31-35
    { const std::string mb = slurp(dir + "/markers.bin"); decode_markers((const uint8_t*)mb.data(), mb.size(), db.params, db.markers); }
    db.sketches.resize(db.markers.size());
    if (exists(dir + "/sketches.db") && exists(dir + "/index.db")) {                 // sketch_db.rs is_consolidated_db
        const std::string ib = slurp(dir + "/index.db");
        const std::vector<IndexEntry> idx = decode_index((const uint8_t*)ib.data(), ib.size());
        if (idx.size() != db.markers.size()) throw std::runtime_error("index.db and markers.bin disagree on the number of sketches");
        const std::string blob = slurp(dir + "/sketches.db");
        for (size_t j = 0; j < idx.size(); j++) {                                    // search.rs:150-158: sketch j of the index belongs to marker sketch j
            if (idx[j].offset > blob.size() || idx[j].length > blob.size() - idx[j].offset) throw std::runtime_error("index.db points outside sketches.db");
            SketchFileParams p; decode_sketch((const uint8_t*)blob.data() + idx[j].offset, idx[j].length, p, db.sketches[j], nullptr);
        }
    } else {
        for (size_t j = 0; j < db.markers.size(); j++) {                             // search.rs:160-170
            std::string path = dir + "/" + base_name(db.markers[j].file_name + ".sketch");
            if (!exists(path)) path = dir + "/" + std::to_string(db.markers[j].contig_order) + "_" + base_name(db.markers[j].file_name + ".sketch");   // -i naming, sketch.rs:70-74
            const std::string bytes = slurp(path);
            SketchFileParams p; decode_sketch((const uint8_t*)bytes.data(), bytes.size(), p, db.sketches[j], nullptr);
        }
    }
    return db;
}

}  // namespace skhost
