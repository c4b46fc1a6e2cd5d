// capi_test.cpp -- extern "C" hooks so the CPU test-suite can exercise the host code (FASTA reader, writers) via ctypes.
#include <cstdlib>
#include <cstring>

#include "host.hpp"

using namespace skhost;

static char* dup(const std::string& s) { char* p = (char*)malloc(s.size() + 1); memcpy(p, s.data(), s.size()); p[s.size()] = 0; return p; }

static std::vector<GenomeInfo> infos(uint32_t n, const char** files, const char** ctg0, const uint32_t* total_len) {
    std::vector<GenomeInfo> g(n);
    for (uint32_t i = 0; i < n; i++) { g[i].file_name = files[i]; g[i].contigs.push_back(ctg0[i]); g[i].contig_lengths.push_back(total_len[i]); }
    return g;
}
static OutOpts opts(uint32_t flags) { OutOpts o; o.ci = flags & 1; o.detailed = flags & 2; o.short_header = flags & 4; o.diagonal = flags & 8; o.full_matrix = flags & 16; o.distance = flags & 32; return o; }

extern "C" {
void skhost_free(char* p) { free(p); }

// returns "name\tlen\n" per kept record (>= min_len) -- checks reader semantics
char* skhost_fasta_summary(const char* path, uint64_t min_len) {
    try {
        std::string s;
        for (auto& r : read_fasta(path)) if (r.seq.size() >= min_len) s += r.name + "\t" + std::to_string(r.seq.size()) + "\n";
        return dup(s);
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
}
