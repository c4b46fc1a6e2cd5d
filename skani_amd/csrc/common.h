// common.h -- constants, hashes and small POD types shared by host and device code.
// Constants restate the reference's src/params.rs (file:line cited per item).
#pragma once
#include <cstdint>

#include "dev.h"

namespace skh {

// ---- reference constants -------------------------------------------------------------------------
constexpr uint32_t K_MARKER = 21;            // params.rs:36 K_MARKER_DNA
constexpr uint32_t CHUNK_SIZE = 20000;       // params.rs:40 CHUNK_SIZE_DNA (fragment_length, params.rs:125-134)
constexpr uint32_t MIN_LENGTH_COVER = 500;   // params.rs:44
constexpr uint32_t BP_CHAIN_BAND = 2500;     // params.rs:45
constexpr int32_t MAX_GAP = 300;             // params.rs:19 D_MAX_GAP_LENGTH
constexpr int32_t MAX_LIN = 5000;            // params.rs:21 D_MAX_LIN_LENGTH
constexpr int32_t ANCHOR_SCORE = 20;         // params.rs:22 D_ANCHOR_SCORE_ANI
constexpr uint32_t MIN_ANCHORS = 3;          // params.rs:24 D_MIN_ANCHORS_ANI
constexpr int32_t MIN_SCORE = 45;            // chain.rs:113  3 * 20 * 0.75
constexpr uint32_t SCREEN_MIN_KMERS = 20;    // params.rs:49
constexpr uint32_t REGRESS_CUTOFF = 150000;  // params.rs:53 TOTAL_BASES_REGRESS_CUTOFF

// ---- seeding geometry -----------------------------------------------------------------------------
constexpr uint32_t SEED_THREADS = 256;
constexpr uint32_t SEED_RUN = 32;                          // windows per thread
constexpr uint32_t SEED_TILE = SEED_THREADS * SEED_RUN;    // 8192 windows per workgroup
constexpr uint32_t CONTIG_ALIGN = 64;                      // contigs start on a 64-base (16 B packed) boundary

// Thomas Wang / minimap2 64-bit mix -- types.rs:86-96 (mm_hash64)
__host__ __device__ __forceinline__ uint64_t mm_hash64(uint64_t key) {
    key = ~(key + (key << 21));
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// table hash for the per-sketch seed tables (internal; murmur3 finaliser)
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// Padded genome coordinates: a seed at (contig c, pos) lives at goff[c] + pos, where the first contig starts at CTG_PAD and
// consecutive contigs are CTG_PAD apart.  A genome whose padded span stays below 2^31 - CTG_PAD keeps them in 32-bit records; a sketch set with a
// longer genome is "wide" (internal.h skh_sketch_set::wide): 64-bit coordinates beside position indices.
// CTG_PAD exceeds every distance the chaining DP can bridge (D_MAX_LIN_LENGTH, BP_CHAIN_BAND), so "same contig" is implied
// by "close enough" and an anchor needs no contig field; the margins at both ends of the coordinate range let the DP fold
// the strand test into the same comparison (chain.hip).  A position is stored as gpos << 1 | canonical-strand bit.
#ifndef SKH_CTG_PAD
#define SKH_CTG_PAD 8192            // (a build parameter so that a test build can put a handful of small contigs 2^32 coordinates apart: tests/emu/build_emu.py, variant "bigpad")
#endif
constexpr uint32_t CTG_PAD = SKH_CTG_PAD;
static_assert(CTG_PAD > (uint32_t)MAX_LIN && CTG_PAD > BP_CHAIN_BAND, "contig padding must exceed the chaining reach");
template <class Arr, class Co>
__host__ __device__ __forceinline__ uint32_t ctg_of(const Arr& goff, uint32_t n_ctg, Co gpos) {   // largest c with goff[c] <= gpos
    uint32_t lo = 0, hi = n_ctg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (goff[mid] <= gpos) lo = mid; else hi = mid; }
    return lo;
}

// A genome's seed table works on table_hash(seed, salt): the salt is 0 unless the genome's seeds crowded one stretch of the hash range under it (a table
// slice's overflow slots ran out, sketch_build.hip) -- such a genome is indexed under the next salt, and the join hashes the enumerated seeds with the
// probed genome's salt.  A bijection of the seed for every salt: equal hash <=> equal seed.
__host__ __device__ __forceinline__ uint32_t table_hash(uint32_t seed, uint32_t salt) { return mix32(seed ^ salt); }

// home slot (bucket) of a hashed seed in a genome's seed table: monotone in the hash, any bucket count
__host__ __device__ __forceinline__ uint32_t seed_bucket(uint32_t hash, uint32_t n_buckets) { return (uint32_t)(((uint64_t)hash * n_buckets) >> 32); }

// contig descriptor inside a packed genome set
struct ContigDesc {
    uint64_t base;      // first base in the packed stream (multiple of CONTIG_ALIGN)
    uint32_t len;       // bases
    uint32_t genome;    // genome id within the set
    uint32_t index;     // contig index within its genome (types.rs:124 contig_index)
    uint32_t has_n;     // set by the pack kernel when the contig contains a masked byte
    uint32_t goff;      // padded-coordinate start of the contig within its genome (CTG_PAD): low word,
    uint32_t goff_hi;   // high word (non-zero only in genomes beyond 2^31 padded bases: "wide" sketch sets, internal.h)
};

// one seeding workgroup's work: SEED_TILE consecutive windows of one contig
struct SeedTile {
    uint32_t contig;    // index into ContigDesc[]
    uint32_t first;     // tile index within the contig: windows i in [20 + first*SEED_TILE, ...)
};

// Seed table of a genome (sketch_build.hip build_tables_kernel): open addressing, n_buckets home slots cut into slices of TAB_SLICE; every slice is
// followed by TAB_SLACK overflow slots of its own, so a probe that starts in a slice ends in it.  Every run of occupied slots ascends by hash: a probe
// walks from its home slot while the slot's hash is smaller than its own (an empty slot is all ones and ends every walk).  Slot = mix32(seed) << 32 | x:
//   x <  TAB_LISTED              the seed occurs once in the genome: x IS its position (padded coordinate << 1 | strand) -- no second request
//   x = TAB_LISTED | code << 29 | offset   2 .. band occurrences (or one beyond 2^31): list storage of the genome at `offset`: count, then the positions
//                                ascending; code = count - 1 for 2..4 occurrences (most lists: the join then needs no look at the list head), else 0
//   x = TAB_REPETITIVE           more than band occurrences: the join drops the seed entirely (chain.rs:694-696)
constexpr uint64_t TAB_EMPTY = ~0ull;
constexpr uint32_t TAB_SLICE_SHIFT = 11, TAB_SLICE = 1u << TAB_SLICE_SHIFT, TAB_SLACK = 128;
constexpr uint32_t TAB_LISTED = 0x80000000u, TAB_REPETITIVE = 0xFFFFFFFDu, TAB_OFF_BITS = 29, TAB_OFF_MASK = (1u << TAB_OFF_BITS) - 1u;   // list offsets stay below TAB_OFF_MASK - 8: no payload equals
                                                                                                                          // TAB_REPETITIVE or all ones (hash 0xFFFFFFFF | all ones would read as an empty slot)
__host__ __device__ __forceinline__ uint32_t tab_list_code(uint32_t x) { return (x >> TAB_OFF_BITS) & 3u; }   // 0: count at the list head, else count - 1
// Occupancy filter of a seed table, staged in LDS by the join: one 32-bit word per TAB_FILTER_HOMES home slots (home >> 4: the slices of the table build own
// whole words), two bits per distinct seed chosen by its low hash bits.  A probe whose two bits are not both set is absent for certain and costs no
// memory request; an absent seed passes with probability (1 - e^(-1/2))^2 = 15 % (one bit per home slot, round 1: 39 %).  4 bits per position: 20 KB per 5 Mbp genome.
#ifndef SKH_TAB_FILTER_SHIFT   // (experiment builds, tools/exp/lib_variant.sh: 5 = a 10 KB filter per 5 Mbp genome, 40 % of the absent seeds pass)
#define SKH_TAB_FILTER_SHIFT 4
#endif
constexpr uint32_t TAB_FILTER_SHIFT = SKH_TAB_FILTER_SHIFT, TAB_FILTER_HOMES = 1u << TAB_FILTER_SHIFT;
__host__ __device__ __forceinline__ uint32_t tab_filter_bits(uint32_t hash) { return (1u << (hash & 31u)) | (1u << ((hash >> 5) & 31u)); }
__host__ __device__ __forceinline__ uint32_t tab_slot(uint32_t home) { return home + (home >> TAB_SLICE_SHIFT) * TAB_SLACK; }   // physical slot of a home slot

// Incidence key of the marker screen (screen.hip): (marker's low 10 bits << 22 | is_query << 21 | genome) << 32 | marker >> 10.  The lists are sorted by
// the LOW 32 bits only -- the marker's leading 16 bases.
constexpr int SCREEN_ID_BITS = 21;                      // genome id field inside the key (a marker is 42 bits)
constexpr uint64_t SCREEN_ID_MASK = (1ull << SCREEN_ID_BITS) - 1;
__host__ __device__ __forceinline__ uint64_t screen_key(uint64_t marker, uint32_t is_query, uint32_t genome) {
    return ((((marker & 0x3FFull) << (SCREEN_ID_BITS + 1)) | ((uint64_t)is_query << SCREEN_ID_BITS) | genome) << 32) | (marker >> 10);
}

}  // namespace skh
