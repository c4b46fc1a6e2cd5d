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

bool parse_fasta_plain(const std::string& path, uint8_t* dst, size_t cap, size_t min_len, size_t* used, std::vector<std::string>& names, std::vector<uint64_t>& lens) {
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
            size_t l = le ? (size_t)(le - data) : n, len = l - p;
            if (memchr(data + p, '\r', len)) { for (size_t i = p; i < l; i++) if (data[i] != '\r') dst[at++] = (uint8_t)data[i]; }
            else { memcpy(dst + at, data + p, len); at += len; }
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
