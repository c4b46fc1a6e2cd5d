// fastx.cpp -- FASTA / FASTQ (.gz) ingest with the record semantics of needletail as used by file_io.rs:158-181.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <cstdio>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "host.hpp"

namespace skhost {

static std::string slurp(const std::string& path) {
    gzFile f = gzopen(path.c_str(), "rb");          // transparently reads plain files too
    if (!f) throw std::runtime_error("cannot open " + path);
    gzbuffer(f, 1 << 20);
    std::string data; std::vector<char> buf(1 << 22);
    for (;;) {
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { gzclose(f); throw std::runtime_error("read error in " + path); }
        if (n == 0) break;
        data.append(buf.data(), (size_t)n);
    }
    gzclose(f);
    return data;
}

std::vector<Record> read_fasta(const std::string& path) {
    const std::string data = slurp(path);
    std::vector<Record> out;
    size_t p = 0, n = data.size();
    while (p < n && (data[p] == '\n' || data[p] == '\r' || data[p] == ' ' || data[p] == '\t')) p++;
    if (p == n) return out;
    if (data[p] == '@') {   // FASTQ (needletail's parse_fastx_file takes both, file_io.rs:158): @name / sequence line(s) / +[name] / quality of the same length
        while (p < n) {
            while (p < n && (data[p] == '\n' || data[p] == '\r')) p++;
            if (p >= n) break;
            if (data[p] != '@') throw std::runtime_error(path + " is not a valid fastq file");
            size_t eol = data.find('\n', p); if (eol == std::string::npos) eol = n;
            Record r; r.name = data.substr(p + 1, eol - p - 1);
            while (!r.name.empty() && r.name.back() == '\r') r.name.pop_back();
            p = eol < n ? eol + 1 : n;
            while (p < n && data[p] != '+') {                                       // sequence lines up to the separator
                eol = data.find('\n', p); if (eol == std::string::npos) eol = n;
                for (size_t i = p; i < eol; i++) if (data[i] != '\r') r.seq.push_back(data[i]);
                p = eol < n ? eol + 1 : n;
            }
            if (p >= n) throw std::runtime_error(path + " is a truncated fastq file");
            eol = data.find('\n', p); p = eol == std::string::npos ? n : eol + 1;    // the '+' line
            size_t q = 0;                                                           // quality: as many characters as the sequence has (may contain '@' and '+')
            while (p < n && q < r.seq.size()) { if (data[p] != '\n' && data[p] != '\r') q++; p++; }
            if (q < r.seq.size()) throw std::runtime_error(path + " is a truncated fastq file");
            out.push_back(std::move(r));
        }
        return out;
    }
    if (data[p] != '>') throw std::runtime_error(path + " is not a valid fasta/fastq file");
    while (p < n) {
        size_t eol = data.find('\n', p);
        if (eol == std::string::npos) eol = n;
        Record r;
        r.name = data.substr(p + 1, eol - p - 1);
        while (!r.name.empty() && r.name.back() == '\r') r.name.pop_back();
        size_t q = eol < n ? eol + 1 : n;
        size_t next = data.find("\n>", eol);
        size_t end = next == std::string::npos ? n : next + 1;
        r.seq.reserve(end - q);
        for (size_t i = q; i < end; i++) { const char ch = data[i]; if (ch != '\n' && ch != '\r') r.seq.push_back(ch); }
        out.push_back(std::move(r));
        p = end;
    }
    return out;
}

// A file of `par_min` bytes or more (a eukaryote's chromosomes in one file) is parsed by `threads` threads: the file is cut at byte offsets, every thread
// takes the lines that START in its stretch -- header lines ('>' at a line start) and sequence lines --, first to count the bases between the
// headers, then, once the contigs' lengths and places are known, to copy them.  One thread, line by line, managed 1.9 GB/s: 2.4 s of the 3.0 s a
// human-sized pair took end to end.
namespace {
struct Stretch {                                     // what one thread found in its part of the file
    struct Header { size_t pos, end; uint64_t bases; };   // the header line [pos, end) and the bases behind it up to the next header of this stretch / its end
    uint64_t before = 0;                             // bases in front of the stretch's first header: they belong to the contig an earlier stretch opened
    std::vector<Header> headers;
};
// calls on_header(pos, end) / on_seq(pos, end) for every line that starts in [lo, hi); `end` excludes the line feed
template <class H, class S>
void walk_lines(const char* data, size_t n, size_t lo, size_t hi, bool lo_starts_a_line, H on_header, S on_seq) {
    size_t p = lo;
    if (!lo_starts_a_line && p > 0 && data[p - 1] != '\n') { const char* nl = (const char*)memchr(data + p, '\n', n - p); p = nl ? (size_t)(nl - data) + 1 : n; }   // the line in progress belongs to the stretch before
    while (p < hi && p < n) {
        const char* le = (const char*)memchr(data + p, '\n', n - p);
        const size_t l = le ? (size_t)(le - data) : n;
        if (data[p] == '>') on_header(p, l); else on_seq(p, l);
        p = l < n ? l + 1 : n;
    }
}
inline size_t copy_line(const char* data, size_t p, size_t l, uint8_t* dst) {          // the line's bytes without carriage returns; returns their number
    const size_t len = l - p;
    if (!memchr(data + p, '\r', len)) { memcpy(dst, data + p, len); return len; }
    size_t k = 0; for (size_t i = p; i < l; i++) if (data[i] != '\r') dst[k++] = (uint8_t)data[i];
    return k;
}
inline size_t count_line(const char* data, size_t p, size_t l) {
    size_t k = l - p; const char* q = data + p; size_t left = l - p;
    while (const char* r = (const char*)memchr(q, '\r', left)) { k--; left -= (size_t)(r - q) + 1; q = r + 1; }
    return k;
}
}  // namespace

bool parse_fasta_plain(const std::string& path, uint8_t* dst, size_t cap, size_t min_len, size_t* used, std::vector<std::string>& names, std::vector<uint64_t>& lens,
                       int threads, size_t par_min) {
    *used = 0; names.clear(); lens.clear();
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); throw std::runtime_error("cannot stat " + path); }
    const size_t n = (size_t)sb.st_size;
    if (n == 0) { close(fd); return true; }
    void* map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) throw std::runtime_error("cannot map " + path);
    struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, n};
    (void)madvise(map, n, MADV_SEQUENTIAL);
    const char* data = (const char*)map;
    if (n >= 2 && (unsigned char)data[0] == 0x1f && (unsigned char)data[1] == 0x8b) return false;        // gzip
    size_t p = 0;
    while (p < n && (data[p] == '\n' || data[p] == '\r' || data[p] == ' ' || data[p] == '\t')) p++;
    if (p == n) return true;
    if (data[p] == '@') return false;                                                                     // FASTQ
    if (data[p] != '>') throw std::runtime_error(path + " is not a valid fasta/fastq file");
    if (cap < n) throw std::runtime_error("parse_fasta_plain: destination smaller than the file");
    if (threads > 1 && n >= par_min) {
        // ---- several threads.  The first header is at p (a line start: only line ends and blanks precede it); the stretches begin there.
        const size_t T = (size_t)threads, first = p;
        std::vector<size_t> cut(T + 1); for (size_t t = 0; t <= T; t++) cut[t] = first + (n - first) / T * t; cut[T] = n;
        std::vector<Stretch> st(T);
        auto each = [&](auto fn) { std::vector<std::thread> th; for (size_t t = 1; t < T; t++) th.emplace_back(fn, t); fn((size_t)0); for (auto& x : th) x.join(); };
        each([&](size_t t) {
            Stretch& S = st[t];
            walk_lines(data, n, cut[t], cut[t + 1], t == 0,
                       [&](size_t a, size_t e) { S.headers.push_back({a, e, 0}); },
                       [&](size_t a, size_t e) { const size_t k = count_line(data, a, e); if (S.headers.empty()) S.before += k; else S.headers.back().bases += k; });
        });
        // contigs in file order: a header's bases run on through the header-less fronts of the stretches behind it
        struct Contig { size_t t, h; uint64_t len, at; bool kept; };
        std::vector<Contig> ctg;
        for (size_t t = 0; t < T; t++) {
            if (!ctg.empty()) ctg.back().len += st[t].before;
            for (size_t h = 0; h < st[t].headers.size(); h++) ctg.push_back({t, h, st[t].headers[h].bases, 0, false});
        }
        uint64_t at = 0;
        for (auto& c : ctg) { c.kept = c.len >= min_len; c.at = at; if (c.kept) at += c.len; }             // file_io.rs:176: shorter contigs are skipped
        // where every stretch writes: its front continues the last contig opened before it, then its own contigs
        std::vector<uint64_t> front_at(T, 0); std::vector<char> front_kept(T, 0);
        {
            size_t ci = 0; uint64_t run = 0; bool open = false, kept = false;                               // `run`: bases of the open contig written by earlier stretches
            for (size_t t = 0; t < T; t++) {
                if (open) { front_at[t] = ctg[ci - 1].at + run; front_kept[t] = kept; run += st[t].before; }
                for (size_t h = 0; h < st[t].headers.size(); h++) { open = true; kept = ctg[ci].kept; run = st[t].headers[h].bases; ci++; }
            }
        }
        each([&](size_t t) {
            size_t ci = 0; for (size_t u = 0; u < t; u++) ci += st[u].headers.size();                         // index of this stretch's first own contig
            uint8_t* w = front_kept[t] ? dst + front_at[t] : nullptr;
            walk_lines(data, n, cut[t], cut[t + 1], t == 0,
                       [&](size_t, size_t) { w = ctg[ci].kept ? dst + ctg[ci].at : nullptr; ci++; },
                       [&](size_t a, size_t e) { if (w) w += copy_line(data, a, e, w); });
        });
        for (auto& c : ctg) if (c.kept) {
            const Stretch::Header& H = st[c.t].headers[c.h];
            std::string name(data + H.pos + 1, H.end - H.pos - 1);
            while (!name.empty() && name.back() == '\r') name.pop_back();
            names.push_back(std::move(name)); lens.push_back(c.len);
        }
        *used = at;
        return true;
    }
    size_t at = 0;                                                                                        // bytes kept so far
    while (p < n) {                                                                                       // data[p] == '>'
        const char* eol = (const char*)memchr(data + p, '\n', n - p);
        const size_t e = eol ? (size_t)(eol - data) : n;
        std::string name(data + p + 1, e - p - 1);
        while (!name.empty() && name.back() == '\r') name.pop_back();
        p = e < n ? e + 1 : n;
        const size_t start = at;
        while (p < n && data[p] != '>') {                                                                 // sequence lines up to the next header line
            const char* le = (const char*)memchr(data + p, '\n', n - p);
            const size_t l = le ? (size_t)(le - data) : n;
            at += copy_line(data, p, l, dst + at);
            p = l < n ? l + 1 : n;
        }
        if (at - start >= min_len) { names.push_back(std::move(name)); lens.push_back(at - start); }
        else at = start;                                                                                  // file_io.rs:176: the contig is skipped
    }
    *used = at;
    return true;
}

LoadedGenomes load_genomes(const std::vector<std::string>& files_in, bool individual_contig, int threads) {
    constexpr size_t MIN_LENGTH_CONTIG = 500;                                   // params.rs:42, file_io.rs:176
    std::vector<std::string> files = files_in;
    struct PerFile { std::vector<Record> recs; bool ok = true; };
    std::vector<PerFile> per(files.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= files.size()) break;
            try {
                auto recs = read_fasta(files[i]);
                for (auto& r : recs) if (r.seq.size() >= MIN_LENGTH_CONTIG) per[i].recs.push_back(std::move(r));
            } catch (const std::exception& e) { per[i].ok = false; fprintf(stderr, "WARN %s; skipping.\n", e.what()); }
        }
    };
    int nt = std::max(1, std::min<int>(threads, (int)files.size()));
    std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(worker); for (auto& t : th) t.join();
    // sketches sorted by (file_name, contig_order): file_io.rs:250 / types.rs:360-364
    std::vector<size_t> order(files.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return files[a] < files[b]; });
    LoadedGenomes lg; lg.contig_off.push_back(0);
    for (size_t i : order) {
        if (!per[i].ok) { lg.skipped.push_back(files[i]); continue; }
        if (per[i].recs.empty()) { fprintf(stderr, "WARN File %s consists of only contigs < 500 bp. Skipping this file.\n", files[i].c_str()); lg.skipped.push_back(files[i]); continue; }
        if (!individual_contig) {
            GenomeInfo gi; gi.file_name = files[i];
            for (auto& r : per[i].recs) {
                gi.contigs.push_back(r.name); gi.contig_lengths.push_back((uint32_t)r.seq.size());
                lg.bases += r.seq; lg.contig_off.push_back(lg.bases.size()); lg.contig_genome.push_back((uint32_t)lg.info.size());
            }
            lg.info.push_back(std::move(gi));
        } else {
            uint64_t j = 0;
            for (auto& r : per[i].recs) {
                GenomeInfo gi; gi.file_name = files[i]; gi.contig_order = j++; gi.contigs.push_back(r.name); gi.contig_lengths.push_back((uint32_t)r.seq.size());
                lg.bases += r.seq; lg.contig_off.push_back(lg.bases.size()); lg.contig_genome.push_back((uint32_t)lg.info.size());
                lg.info.push_back(std::move(gi));
            }
        }
    }
    return lg;
}

}  // namespace skhost
