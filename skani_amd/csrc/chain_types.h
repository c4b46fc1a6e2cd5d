// chain_types.h -- PairDesc, Chunk, Interval: the records the chaining kernels pass to each other.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ views & descriptors
// One record per genome pair.  It carries direct pointers to the two sketches' arrays (already advanced to the genome's first
// element), so the pairs of one call may draw their sketches from any number of resident sketch sets (a sharded database).
struct PairDesc {
    // A = enumerated sketch (position order); p_g = padded coordinate << 1 | canonical
    const uint32_t *a_seed, *a_g; const uint32_t* a_rep;   // a_rep: the set's "repetitive seed" bits; bit a_pos0 + i belongs to position i
    // B = probed sketch: seed table (common.h: position or list reference in the slot), list storage, bucket-occupancy bitmap
    const uint32_t* b_ms; const uint64_t* b_tab; const uint32_t* b_bmap;
    const uint32_t *a_goff, *b_goff;   // padded contig starts (common.h CTG_PAD), a_nctg + 1 / b_nctg + 1 entries
    uint32_t a_n;       // positions in A
    uint32_t a_pos0;    // A's first position in its set's position numbering
    uint32_t b_nbk;     // B: buckets (home slots) of its seed table
    uint32_t b_salt;    // B's table works on table_hash(seed, b_salt) (common.h): the join hashes A's seeds with it
    uint32_t flags;     // bit2: switched (chain.rs:649)
    uint32_t tile0;     // first join tile of this pair (global over the call)
    uint32_t a_nctg, b_nctg;
    // finalisation inputs (ref/query in the caller's sense, NOT A/B)
    uint32_t nctg_q, nctg_r;
    uint64_t ref_total_len, query_total_len;
    float q10_q, q50_q, q90_q, q10_r, q50_r, q90_r;
};

// What a pair's descriptor takes from ONE genome, resident on the device: one table per sketch set, uploaded once (chain.hip dev_halves).  A chaining call then
// uploads 16 bytes per pair (PairRec) and expand_pairs_kernel makes the descriptors from the two tables -- the host used to fill 136 bytes per pair and copy them over
// PCIe (1.3 MB for the headline's 9,500 pairs, in front of the join on the step's critical path).
struct GenomeDev {
    const uint32_t *seed, *g, *rep, *ms, *bmap, *goff; const uint64_t* tab;
    uint32_t n_pos, pos0, nbk, salt, nctg, pad;
    uint64_t total_len; float q10, q50, q90, pad2;
};
// flags: bit 2 switched (chain.rs:649); bit 3 a genome without contigs (chain.rs:618-620: no tiles); bits 8..19 the reference's set, 20..31 the query's set
struct PairRec { uint32_t r, q, flags, tile0; };
constexpr uint32_t PAIR_MAX_SETS = 4096;

// ------------------------------------------------------------------------------------------------ coordinate width
// A run over pairs of ordinary sketch sets keeps every coordinate in 32 bits (Narrow).  A run that involves a WIDE set (internal.h: a genome beyond
// 2^31 padded bases) works on 64-bit coordinates from the chunking on (Wide): the join is the same -- a wide set's position records and table
// payloads are position indices --, widen_anchors_kernel then turns the anchors into 64-bit coordinates, and the kernels behind it are instantiated
// a second time.  The sides of a pair may differ (a 3 Gbp genome against a bacterial one): CoArr reads either record width.
struct CoArr {
    const void* p; uint32_t is64;
    __host__ __device__ __forceinline__ uint64_t operator[](uint64_t i) const { return is64 ? ((const uint64_t*)p)[i] : (uint64_t)((const uint32_t*)p)[i]; }
};
struct WidePair {                // per pair of a Wide run, beside its PairDesc
    CoArr a_g;                   // A's positions as coordinate << 1 | canonical (PairDesc::a_g holds indices when A's set is wide)
    CoArr a_goff, b_goff;        // padded contig starts
    const uint64_t* b_g64;       // B's coordinate records when B's set is wide (its anchors carry indices), else null
    uint32_t a_is_index;         // A's set is wide: anc_q from the join is a position index
    uint32_t pad;
};
struct Narrow {   // (Arr: pointers out of the pair record are named as what they are, global memory -- dev.h global_of)
    using Co = uint32_t; using Arr = GlobalPtr<uint32_t>; static constexpr bool wide = false;
    static __device__ __forceinline__ Arr a_g(const PairDesc& pd, const WidePair*, uint32_t) { return global_of(pd.a_g); }
    static __device__ __forceinline__ Arr a_goff(const PairDesc& pd, const WidePair*, uint32_t) { return global_of(pd.a_goff); }
    static __device__ __forceinline__ Arr b_goff(const PairDesc& pd, const WidePair*, uint32_t) { return global_of(pd.b_goff); }
};
struct Wide {
    using Co = uint64_t; using Arr = CoArr; static constexpr bool wide = true;
    static __device__ __forceinline__ Arr a_g(const PairDesc&, const WidePair* wp, uint32_t p) { return wp[p].a_g; }
    static __device__ __forceinline__ Arr a_goff(const PairDesc&, const WidePair* wp, uint32_t p) { return wp[p].a_goff; }
    static __device__ __forceinline__ Arr b_goff(const PairDesc&, const WidePair* wp, uint32_t p) { return wp[p].b_goff; }
};

constexpr uint32_t JOIN_TILE = 1024;    // positions per join tile (one wave: 4 rounds x 4 probes per lane)
constexpr uint32_t JOIN_GROUP = 4;      // tiles per join workgroup (one wave each); count pass with 2 / 4 / 8 / 16: 2.92 / 2.16 / 2.42 / 2.60 ms
constexpr uint32_t JOIN_THREADS = 64 * JOIN_GROUP;
constexpr uint32_t NONE = 0xFFFFFFFFu;

// An anchor is 8 bytes in two arrays: anc_q = padded query coordinate, anc_r = padded ref coordinate << 1 | reverse_match
// (chunking only needs the first).  Contigs are recovered from the padded contig-start tables where a stage needs them
// (chunk boundaries, interval records).
struct Chunk { uint32_t a_begin, a_end, s_begin, s_end, qoff, qctg; };   // batch-relative anchor / seed-list ranges; the chunk's query contig and its padded start (Wide runs: qoff is not used, the start is a_goff[qctg])
struct Interval { uint32_t score, na, q0, q1, r0, r1, rctg, qctg, chunk, rev; };   // types.rs:508-519 field order = sort order
