// chain_types.h -- PairDesc, Chunk, Interval: the records the chaining kernels pass to each other.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ views & descriptors
// One record per genome pair.  It carries direct pointers to the two sketches' arrays (already advanced to the genome's first
// element), so the pairs of one call may draw their sketches from any number of resident sketch sets (a sharded database).
struct PairDesc {
    // A = enumerated sketch (position order); p_g = padded coordinate << 1 | canonical
    const uint32_t *a_hash, *a_g; const uint32_t* a_rep;   // a_hash = mix32(seed); a_rep: the set's "repetitive seed" bits; bit a_pos0 + i belongs to position i
    // B = probed sketch: seed table (common.h: position or list reference in the slot), list storage, bucket-occupancy bitmap
    const uint32_t* b_ms; const uint64_t* b_tab; const uint32_t* b_bmap;
    const uint32_t *a_goff, *b_goff;   // padded contig starts (common.h CTG_PAD), a_nctg + 1 / b_nctg + 1 entries
    uint32_t a_n;       // positions in A
    uint32_t a_pos0;    // A's first position in its set's position numbering
    uint32_t b_nbk;     // B: buckets (home slots) of its seed table
    uint32_t flags;     // bit2: switched (chain.rs:649)
    uint32_t tile0;     // first join tile of this pair (global over the call)
    uint32_t a_nctg, b_nctg;
    // finalisation inputs (ref/query in the caller's sense, NOT A/B)
    uint32_t nctg_q, nctg_r;
    uint64_t ref_total_len, query_total_len;
    float q10_q, q50_q, q90_q, q10_r, q50_r, q90_r;
};

constexpr uint32_t JOIN_TILE = 1024;    // positions per join tile (one wave: 4 rounds x 4 probes per lane)
constexpr uint32_t JOIN_GROUP = 4;      // tiles per join workgroup (one wave each)
constexpr uint32_t NONE = 0xFFFFFFFFu;

// An anchor is 8 bytes in two arrays: anc_q = padded query coordinate, anc_r = padded ref coordinate << 1 | reverse_match
// (chunking only needs the first).  Contigs are recovered from the padded contig-start tables where a stage needs them
// (chunk boundaries, interval records).
struct Chunk { uint32_t a_begin, a_end, s_begin, s_end, qoff, qctg; };   // batch-relative anchor / seed-list ranges; the chunk's query contig and its padded start
struct Interval { uint32_t score, na, q0, q1, r0, r1, rctg, qctg, chunk, rev; };   // types.rs:508-519 field order = sort order
