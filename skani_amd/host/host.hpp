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

// The same for a plain (uncompressed) FASTA file, without the intermediate strings: the file is mapped and the sequence bytes of every contig of at least
// min_len bases are written one after the other to dst (which must hold the file's size); names / lens describe the kept contigs in file order.
// Returns false -- nothing written -- when the file is gzipped or FASTQ: the caller takes read_fasta.  Throws like read_fasta.
bool parse_fasta_plain(const std::string& path, uint8_t* dst, size_t cap, size_t min_len, size_t* used, std::vector<std::string>& names, std::vector<uint64_t>& lens,
                       int threads = 1, size_t par_min = (size_t)256 << 20);   // files of par_min bytes or more are parsed by `threads` threads

// One entry per Sketch (types.rs:252-277): what the host keeps next to the device-resident sketch set.
struct GenomeInfo {
    std::string file_name;
    std::vector<std::string> contigs;      // names of kept contigs
    std::vector<uint32_t> contig_lengths;
    uint64_t contig_order = 0;             // -i mode: index of the contig within its file (file_io.rs:336)
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

// ---- on-disk sketch formats (formats.cpp) ----
struct SeedRecord { uint32_t seed, pos, ctgcanon; };            // ctgcanon = contig_index << 1 | canonical (types.rs:124-143)
struct SketchFileParams { uint64_t c = 125, k = 15, marker_c = 1000; bool use_syncs = false, use_aa = false; };   // params.rs:136-146
struct SketchBlob {                                              // types.rs:252-277
    std::string file_name;
    bool has_seeds = true;                                       // false = markers-only sketch (types.rs:322-340)
    std::vector<SeedRecord> records;                             // (contig, pos) order
    std::vector<std::string> contigs;
    uint64_t total_sequence_length = 0;
    std::vector<uint32_t> contig_lengths;
    uint64_t repetitive_kmers = 0;                               // 0 in every sketch skani writes (types.rs Default)
    std::vector<uint64_t> markers;                               // ascending
    uint64_t marker_c = 0, c = 0, k = 0, contig_order = 0;
    bool individual_contig = false, amino_acid = false;
};
struct IndexEntry { std::string file_name; uint64_t offset, length; };    // sketch_db.rs:10-15
struct SketchDb { SketchFileParams params; std::vector<SketchBlob> markers, sketches; };   // sketches[j] belongs to markers[j]

std::string encode_sketch(const SketchFileParams&, const SketchBlob&);                               // v0.3 (SketchParams, Sketch)
// v0.3 layout, falling back to the pre-0.3 layout; *format = 3 or 2.  The whole buffer must be one blob.  Returns bytes used.
size_t decode_sketch(const uint8_t* bytes, size_t n, SketchFileParams&, SketchBlob&, int* format);
std::string encode_markers(const SketchFileParams&, const std::vector<SketchBlob>&);                 // markers.bin
void decode_markers(const uint8_t* bytes, size_t n, SketchFileParams&, std::vector<SketchBlob>&);
std::string encode_index(const std::vector<IndexEntry>&);                                            // index.db
std::vector<IndexEntry> decode_index(const uint8_t* bytes, size_t n);
// `skani sketch -o dir` output: sketches.db + index.db + markers.bin, or one .sketch per genome + markers.bin (sketch.rs:13-175)
void write_sketch_db(const std::string& dir, const SketchFileParams&, const std::vector<SketchBlob>&, bool separate_files, bool individual_contig);
// sketches_from_sketch (file_io.rs:680-729): skips markers.bin, reports and skips unreadable files, sorts by file_name
std::vector<SketchBlob> read_sketch_files(const std::vector<std::string>& files, SketchFileParams&);
// a `skani sketch` output folder (either flavour), all sketches loaded (search.rs:16-100 without the lazy fetch)
SketchDb read_sketch_db(const std::string& dir_or_marker_file);

// ---- the GPUs of one node (node.cpp): `triangle --gpus N` forks one process per GPU; the ranks meet in shared memory ----
struct Node;
Node* node_create(int world);                                   // before anything touches the HIP runtime
int node_launch(Node*);                                         // forks the ranks; returns in every rank with its number; the launcher waits for them and exits
int node_rank(const Node*);
int node_world(const Node*);
bool node_barrier(Node*);                                       // false: some rank failed (every collective below likewise)
bool node_all_gather(Node*, const void* send, void* recv, uint64_t bytes);
bool node_all_to_all_v(Node*, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt, const uint64_t* recv_off);
bool node_all_gather_v(Node*, const std::string& mine, std::vector<std::string>& all);   // ragged: every rank's bytes, by rank
bool node_claim_failure(Node*);                                 // marks the run as failed (ends every collective); true for the first caller only: its message is the run's error
bool node_failed(const Node*);
skh_host_collectives node_collectives(Node*);                   // the same collectives as the library's host transport (skh_comm_create_host)

}  // namespace skhost
