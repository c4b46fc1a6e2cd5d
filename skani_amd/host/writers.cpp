// writers.cpp -- the reference's text outputs (file_io.rs:15-139, 364-678), byte for byte.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <sstream>

#include "host.hpp"

namespace skhost {

static std::string f2(float v) { char b[64]; snprintf(b, sizeof b, "%.2f", (double)v); return b; }        // Rust {:.2} on f32
static std::string f0(float v) { char b[64]; snprintf(b, sizeof b, "%.0f", (double)v); return b; }        // Rust {:0} on an integer-valued f32
static std::string trunc_name(const std::string& s, bool short_header) {                                    // types.rs:194-201
    if (!short_header) return s;
    size_t a = s.find_first_not_of(" \t\n\r\f\v"); if (a == std::string::npos) return s;
    size_t b = s.find_first_of(" \t\n\r\f\v", a);
    return s.substr(a, b == std::string::npos ? std::string::npos : b - a);
}

std::string format_header(bool ci, bool detailed) {   // file_io.rs:15-23
    if (!ci && !detailed) return "Ref_file\tQuery_file\tANI\tAlign_fraction_ref\tAlign_fraction_query\tRef_name\tQuery_name\n";
    if (!detailed) return "Ref_file\tQuery_file\tANI\tAlign_fraction_ref\tAlign_fraction_query\tRef_name\tQuery_name\tANI_5_percentile\tANI_95_percentile\n";
    return "Ref_file\tQuery_file\tANI\tAlign_fraction_ref\tAlign_fraction_query\tRef_name\tQuery_name\tNum_ref_contigs\tNum_query_contigs\tANI_5_percentile\t"
           "ANI_95_percentile\tStandard_deviation\tRef_90_ctg_len\tRef_50_ctg_len\tRef_10_ctg_len\tQuery_90_ctg_len\tQuery_50_ctg_len\tQuery_10_ctg_len\t"
           "Avg_chain_len\tTotal_bases_covered\n";
}

std::string format_result(const GenomeInfo& ref, const GenomeInfo& query, const skh_ani_result& r, const OutOpts& o) {   // file_io.rs:84-139
    std::string s = ref.file_name + "\t" + query.file_name + "\t" + f2(r.ani * 100.f) + "\t" + f2(r.af_ref * 100.f) + "\t" + f2(r.af_query * 100.f) + "\t" +
                    trunc_name(ref.contigs[0], o.short_header) + "\t" + trunc_name(query.contigs[0], o.short_header);
    if (o.detailed) {
        s += "\t" + std::to_string(r.num_contigs_r) + "\t" + std::to_string(r.num_contigs_q) + "\t" + f2(r.ci_lower * 100.f) + "\t" + f2(r.ci_upper * 100.f) + "\t" +
             f2(r.std * 100.f) + "\t" + f0(r.q90_r) + "\t" + f0(r.q50_r) + "\t" + f0(r.q10_r) + "\t" + f0(r.q90_q) + "\t" + f0(r.q50_q) + "\t" + f0(r.q10_q) + "\t" +
             std::to_string(r.avg_chain_int_len) + "\t" + std::to_string(r.total_bases_covered);
    } else if (o.ci) {
        s += "\t" + f2(r.ci_lower * 100.f) + "\t" + f2(r.ci_upper * 100.f);
    }
    return s + "\n";
}

static std::string format_perfect(const GenomeInfo& g, const OutOpts& o) {   // file_io.rs:25-82 (integer 100 formatted with {:.2} prints "100")
    std::string n = trunc_name(g.contigs[0], o.short_header);
    std::string s = g.file_name + "\t" + g.file_name + "\t100\t100\t100\t" + n + "\t" + n;
    if (o.detailed) {
        uint64_t tot = 0; for (auto l : g.contig_lengths) tot += l;
        s += "\t" + std::to_string(g.contigs.size()) + "\t" + std::to_string(g.contigs.size()) + "\t100\t100\t0\t-1\t-1\t-1\t-1\t-1\t-1\t0\t" + std::to_string(tot);
    } else if (o.ci) s += "\t100\t100";
    return s + "\n";
}

static bool usable(const skh_ani_result& r) { return !(r.ani == -1.f || std::isnan(r.ani)); }

void format_phylip(const std::vector<GenomeInfo>& g, const std::vector<PairResult>& res, bool use_contig_names, const OutOpts& o,
                   std::string& ani_txt, std::string& af_txt) {
    const size_t n = g.size();
    // anis[x][y], x < y (a later result for the same pair replaces an earlier one, as inserting into the reference's map does); per genome the
    // results it takes part in, so that a row is laid out from a table of n cells instead of n look-ups (the two matrices of 1,000 genomes are
    // 1.5 million cells, nearly all of them the "nothing" value: their text is made once)
    std::vector<std::vector<std::pair<uint32_t, const skh_ani_result*>>> of(n);
    for (auto& p : res) { if (p.ref == p.query) continue; of[p.ref].push_back({p.query, &p.r}); of[p.query].push_back({p.ref, &p.r}); }
    const float perfect = o.distance ? 0.f : 100.f, none = 100.f - perfect;   // file_io.rs:374-375
    const std::string t_perfect = "\t" + f2(perfect), t_none = "\t" + f2(none), t_100 = "\t" + f2(100.f), t_0 = "\t" + f2(0.f);
    std::string a = std::to_string(n) + "\n", f = std::to_string(n) + "\n";
    a.reserve(n * 64 + res.size() * 8 + n * n * 3); f.reserve(n * 64 + n * n * 6);
    std::vector<const skh_ani_result*> cell(n, nullptr);
    char buf[64];
    auto put = [&](std::string& out, float v) { const int len = snprintf(buf, sizeof buf, "\t%.2f", (double)v); out.append(buf, (size_t)len); };
    for (size_t i = 0; i < n; i++) {
        for (auto& e : of[i]) cell[e.first] = usable(*e.second) ? e.second : nullptr;
        const std::string& name = use_contig_names ? g[i].contigs[0] : g[i].file_name;
        a += name; f += name;
        const size_t end = o.full_matrix ? n : (o.diagonal ? i + 1 : i);
        for (size_t j = 0; j < end; j++) {
            if (j == i) { a += t_perfect; continue; }
            const skh_ani_result* r = cell[j];
            if (!r) a += t_none;
            else { const float val = r->ani * 100.f; put(a, o.distance ? 100.f - val : val); }
        }
        a += "\n";
        for (size_t j = 0; j < n; j++) {                                       // the AF matrix is always full (file_io.rs:428-461)
            if (i == j) { f += t_100; continue; }
            const skh_ani_result* r = cell[j];
            if (!r) f += t_0;
            else put(f, (j > i ? r->af_ref : r->af_query) * 100.f);
        }
        f += "\n";
        for (auto& e : of[i]) cell[e.first] = nullptr;
    }
    ani_txt = std::move(a); af_txt = std::move(f);
}

std::string format_sparse(const std::vector<GenomeInfo>& g, const std::vector<PairResult>& res, const OutOpts& o) {
    std::string s = format_header(o.ci, o.detailed);
    if (o.diagonal) for (auto& x : g) s += format_perfect(x, o);
    for (auto& p : res) if (usable(p.r)) s += format_result(g[p.ref], g[p.query], p.r, o);     // row order is unspecified in the reference (hash maps)
    return s;
}

std::string format_query_ref_list(const std::vector<GenomeInfo>& refs, const std::vector<GenomeInfo>& queries, const std::vector<PairResult>& res,
                                  size_t n_max, const OutOpts& o) {
    std::map<std::string, std::vector<const PairResult*>> by_query;            // keyed by query contig name, sorted (file_io.rs:621-637)
    for (auto& p : res) { if (p.r.ani < 0.f || std::isnan(p.r.ani)) continue; by_query[queries[p.query].contigs[0]].push_back(&p); }
    std::string s = format_header(o.ci, o.detailed);
    for (auto& kv : by_query) {
        auto v = kv.second;
        std::stable_sort(v.begin(), v.end(), [](const PairResult* a, const PairResult* b) { return a->r.ani > b->r.ani; });
        for (size_t i = 0; i < std::min(n_max, v.size()); i++) s += format_result(refs[v[i]->ref], queries[v[i]->query], v[i]->r, o);
    }
    return s;
}

}  // namespace skhost
