// host.hpp -- C++ host side above the C ABI (the reference's host is compiled Rust; no Rust toolchain in this image).
// Mirrors the reference's drivers for the hot path's callers: file_io.rs (FASTA ingest, text writers),
// triangle.rs and dist.rs.  Compute goes exclusively through include/skani_hip.h.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/skani_hip.h"

namespace skhost {

struct Record { std::string name; std::string seq; };

// needletail semantics (file_io.rs:158-175): id = header line without '>', sequence without line breaks, no case change;
// gzip detected by magic.  Throws std::runtime_error on unreadable / non-FASTA input.
std::vector<Record> read_fasta(const std::string& path);

// One entry per Sketch (types.rs:252-277): what the host keeps next to the device-resident sketch set.
struct GenomeInfo {
    std::string file_name;
    std::vector<std::string> contigs;      // names of kept contigs
    std::vector<uint32_t> contig_lengths;
};

struct LoadedGenomes {
    std::vector<GenomeInfo> info;           // sorted by (file_name, contig_order) like file_io.rs:250
    std::string bases;                      // concatenated kept contigs
    std::vector<uint64_t> contig_off;       // n_contigs + 1
    std::vector<uint32_t> contig_genome;
    std::vector<std::string> skipped;       // files dropped with a warning
};
// file_io.rs:141-252 (one sketch per file) / :253-362 (-i: one sketch per contig); contigs < 500 bp are skipped.
LoadedGenomes load_genomes(const std::vector<std::string>& files, bool individual_contig, int threads);

struct PairResult { uint32_t ref, query; skh_ani_result r; };   // indices into the ref / query GenomeInfo vectors

struct OutOpts { bool ci = false, detailed = false, short_header = false, diagonal = false, full_matrix = false, distance = false; };

// file_io.rs:15-23, :84-139
std::string format_header(bool ci, bool detailed);
std::string format_result(const GenomeInfo& ref, const GenomeInfo& query, const skh_ani_result& r, const OutOpts& o);
// file_io.rs:364-539: returns the ANI matrix text and the aligned-fraction matrix text
void format_phylip(const std::vector<GenomeInfo>& g, const std::vector<PairResult>& res, bool use_contig_names, const OutOpts& o,
                   std::string& ani_txt, std::string& af_txt);
// file_io.rs:541-606
std::string format_sparse(const std::vector<GenomeInfo>& g, const std::vector<PairResult>& res, const OutOpts& o);
// file_io.rs:608-678
std::string format_query_ref_list(const std::vector<GenomeInfo>& refs, const std::vector<GenomeInfo>& queries, const std::vector<PairResult>& res,
                                  size_t n_max, const OutOpts& o);

}  // namespace skhost
