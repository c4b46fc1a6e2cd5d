/*
 * skani_oracle.cpp -- CPU ORACLE: restatement of the reference skani v0.3.0 hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see skani_oracle.h).  Written from the reference's behaviour as
 * analysed in SURVEY.md section 8; citations "file:line" point into the reference checkout.
 * Third-party crates whose source is not vendored in the reference (partitions 0.2.4 fork,
 * bio 1.4.0 IntervalTree, intervallum 1.4.0, fastrand 1.9.0, gbdt 0.1.1) are restated from
 * their published behaviour and anchored on the reference's call sites.
 */
#include "skani_oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#ifdef __AVX2__
#include <immintrin.h>
#endif
#include <malloc.h>

namespace {

// ---------------------------------------------------------------- constants (params.rs:4-62)
constexpr uint32_t K_MARKER_DNA = 21;           // params.rs:36
constexpr double D_MAX_GAP_LENGTH = 300.;       // params.rs:19
constexpr double D_MAX_LIN_LENGTH = 5000.;      // params.rs:21
constexpr double D_ANCHOR_SCORE_ANI = 20.;      // params.rs:22
constexpr size_t D_MIN_ANCHORS_ANI = 3;         // params.rs:24
constexpr uint32_t CHUNK_SIZE_DNA = 20000;      // params.rs:40
constexpr uint32_t MIN_LENGTH_COVER = 500;      // params.rs:44
constexpr size_t BP_CHAIN_BAND = 2500;          // params.rs:45
constexpr size_t SCREEN_MINIMUM_KMERS = 20;     // params.rs:49
constexpr float OVERLAP_ORTHOLOGOUS_FRACTION = 0.50f;  // params.rs:52
constexpr uint32_t TOTAL_BASES_REGRESS_CUTOFF = 150000;  // params.rs:53

// types.rs:40-49 BYTE_TO_SEQ: A/a=0 C/c=1 G/g=2 T/t/U/u=3, everything else (incl. N) = 0.
struct Lut {
    uint8_t t[256];
    Lut() {
        memset(t, 0, sizeof t);
        t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['U'] = t['u'] = 3;
        t[1] = 1; t[2] = 2; t[3] = 3;  // the table's first row is 0,1,2,3 (types.rs:41)
    }
};
const Lut LUT;

// types.rs:86-96
inline uint64_t mm_hash64(uint64_t key) {
    key = ~(key + (key << 21));
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// ---------------------------------------------------------------- containers
struct SeedPos { uint32_t pos, cc; };  // types.rs:124-143 (cc = contig<<1 | canonical)

// open-addressing u32 -> u64 map hashed with the minimap2 mix (types.rs:98-109, 393-407)
struct SeedMap {
    std::vector<uint32_t> keys; std::vector<uint64_t> vals; size_t n = 0, mask = 0;
    static constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    void init(size_t cap) { size_t c = 16; while (c < cap * 2) c <<= 1; keys.assign(c, EMPTY); vals.assign(c, 0); mask = c - 1; n = 0; }
    void grow() {
        std::vector<uint32_t> ok; ok.swap(keys); std::vector<uint64_t> ov; ov.swap(vals);
        init(ok.size());
        for (size_t i = 0; i < ok.size(); i++) if (ok[i] != EMPTY) *slot(ok[i]) = ov[i];
    }
    uint64_t* slot(uint32_t key) {  // insert-or-find
        if (keys.empty()) init(16);
        if ((n + 1) * 2 > keys.size()) grow();
        size_t h = mm_hash64(key) & mask;
        while (keys[h] != EMPTY && keys[h] != key) h = (h + 1) & mask;
        if (keys[h] == EMPTY) { keys[h] = key; vals[h] = 0; n++; }
        return &vals[h];
    }
    const uint64_t* find(uint32_t key) const {
        if (keys.empty()) return nullptr;
        size_t h = mm_hash64(key) & mask;
        while (keys[h] != EMPTY) { if (keys[h] == key) return &vals[h]; h = (h + 1) & mask; }
        return nullptr;
    }
};

struct MarkerSet {  // HashSet<u64> with the same mix (types.rs:68, 415-427)
    std::vector<uint64_t> keys; size_t n = 0, mask = 0;
    static constexpr uint64_t EMPTY = ~0ull;
    void init(size_t cap) { size_t c = 16; while (c < cap * 2) c <<= 1; keys.assign(c, EMPTY); mask = c - 1; n = 0; }
    void insert(uint64_t k) {
        if (keys.empty()) init(16);
        if ((n + 1) * 2 > keys.size()) { std::vector<uint64_t> ok; ok.swap(keys); init(ok.size()); for (auto x : ok) if (x != EMPTY) insert(x); }
        size_t h = mm_hash64(k) & mask;
        while (keys[h] != EMPTY && keys[h] != k) h = (h + 1) & mask;
        if (keys[h] == EMPTY) { keys[h] = k; n++; }
    }
    bool contains(uint64_t k) const {
        if (keys.empty()) return false;
        size_t h = mm_hash64(k) & mask;
        while (keys[h] != EMPTY) { if (keys[h] == k) return true; h = (h + 1) & mask; }
        return false;
    }
    size_t size() const { return n; }
    template <class F> void for_each(F f) const { for (auto x : keys) if (x != EMPTY) f(x); }
};

}  // namespace

// types.rs:252-277 Sketch, with the v0.3 tagged-index seed table (types.rs:207-244, 281-320)
struct ora_sketch {
    std::string file_name;
    uint32_t c = 0, k = 0, marker_c = 0;
    SeedMap seeds;                               // seed -> tagged (bit0=1: single packed; else index into multi)
    std::vector<std::vector<SeedPos>> multi;
    std::vector<uint32_t> contig_lengths;
    uint64_t total_len = 0;
    MarkerSet markers;
    uint64_t n_positions = 0;

    void add_seed_position(uint32_t seed, SeedPos p) {  // types.rs:281-304
        uint64_t* t = seeds.slot(seed);
        if (*t == 0) {  // fresh (a tagged single always has bit0 set, a multi index is stored +1 below)
            *t = 1ull | ((((uint64_t)p.pos << 31) | p.cc) << 1);
        } else if (*t & 1ull) {
            uint64_t packed = *t >> 1;
            SeedPos e{(uint32_t)(packed >> 31), (uint32_t)(packed & 0x7FFFFFFFu)};
            multi.push_back({e, p});
            *t = (uint64_t)(multi.size()) << 1;  // index+1 so that 0 stays "empty"
        } else {
            multi[(*t >> 1) - 1].push_back(p);
        }
        n_positions++;
    }
    // types.rs:307-320; returns count and fills ptr (single positions are materialised in tmp)
    size_t get(uint32_t seed, const SeedPos** out, SeedPos* tmp) const {
        const uint64_t* t = seeds.find(seed);
        if (!t) return 0;
        if (*t & 1ull) { uint64_t packed = *t >> 1; tmp->pos = (uint32_t)(packed >> 31); tmp->cc = (uint32_t)(packed & 0x7FFFFFFFu); *out = tmp; return 1; }
        const auto& v = multi[(*t >> 1) - 1]; *out = v.data(); return v.size();
    }
    // the same for the seed in slot h of the key table (an enumeration of the map needs no second hash look-up)
    size_t get_at(size_t h, const SeedPos** out, SeedPos* tmp) const {
        const uint64_t t = seeds.vals[h];
        if (t & 1ull) { uint64_t packed = t >> 1; tmp->pos = (uint32_t)(packed >> 31); tmp->cc = (uint32_t)(packed & 0x7FFFFFFFu); *out = tmp; return 1; }
        const auto& v = multi[(t >> 1) - 1]; *out = v.data(); return v.size();
    }
};

struct ora_model {  // gbdt 0.1.1 GBDT restricted to what regression.rs uses
    uint32_t n_trees = 0, n_feat = 0; float shrinkage = 0, bias = 0;
    std::vector<uint32_t> off;
    struct Node { int32_t feat; float thr, pred; int32_t left, right; };
    std::vector<Node> nodes;
};

namespace {

// ---------------------------------------------------------------- seeding
// seeding.rs:225-323
void fmh_seeds_scalar(const uint8_t* s, uint64_t len, uint32_t c, uint32_t k, uint32_t marker_c, uint32_t contig, ora_sketch& sk) {
    const uint32_t marker_k = K_MARKER_DNA;
    if (len < 2 * marker_k) return;                                    // :242
    const uint64_t seed_mask = ~0ull >> (64 - 2 * k);                  // :249
    const uint32_t rshift = 2 * (marker_k - 1);                        // :251
    const uint64_t marker_mask = ~0ull >> (64 - 2 * marker_k);         // :252
    const uint64_t marker_rev_mask = ~(3ull << (2 * marker_k - 2));    // :253
    const uint64_t threshold = ~0ull / (uint64_t)c;                    // :258
    const uint64_t threshold_marker = ~0ull / (uint64_t)marker_c;      // :259
    uint64_t f = 0, r = 0;
    for (uint32_t i = 0; i < marker_k - 1; i++) {                      // :260-269
        uint64_t nf = LUT.t[s[i]], nr = 3 - nf;
        f <<= 2; f |= nf; r >>= 2; r |= nr << rshift;
    }
    uint64_t resume = 0;
    for (uint64_t i = marker_k - 1; i < len; i++) {                    // :271
        uint8_t b = s[i];
        if (b == 78 || b == 110) resume = i + k;                       // :272-275
        uint64_t nf = LUT.t[b], nr = 3 - nf;
        f <<= 2; f |= nf; f &= marker_mask;
        r >>= 2; r &= marker_rev_mask; r |= nr << rshift;
        uint64_t fs = f & seed_mask, rs = r & seed_mask;               // :288-289
        bool canon = fs < rs;                                          // :290
        uint64_t seed = canon ? fs : rs;
        uint64_t h = mm_hash64(seed);
        if (h < threshold && resume <= i) {                            // :300
            sk.add_seed_position((uint32_t)seed, SeedPos{(uint32_t)i, (contig << 1) | (canon ? 1u : 0u)});
            if (h < threshold_marker) sk.markers.insert(f < r ? f : r);  // :311-319
        }
    }
}

// avx2_seeding.rs:33-272 -- what x86-64+AVX2 hosts execute (file_io.rs:194-206).  Four lanes over
// substrings [l*len4, (l+1)*len4+20); tail windows dropped; only 'N' detected; resume = i+21; lane-local.
void fmh_seeds_avx2_semantics(const uint8_t* s, uint64_t slen, uint32_t c, uint32_t k, uint32_t marker_c, uint32_t contig, ora_sketch& sk) {
    const uint32_t marker_k = K_MARKER_DNA;
    const uint64_t len = (slen - marker_k + 1) / 4;                    // :48
    if (slen < 2 * marker_k) return;                                   // :56
    const uint64_t seed_mask = ~0ull >> (64 - 2 * k);
    const uint64_t marker_mask = ~0ull >> (64 - 2 * marker_k);
    const uint64_t rev_marker_mask = ~(3ull << (2 * marker_k - 2));
    const uint64_t threshold = ~0ull / (uint64_t)c, threshold_marker = ~0ull / (uint64_t)marker_c;
    const uint8_t* str[4] = {s, s + len, s + 2 * len, s + 3 * len};    // :49-52
    uint64_t f[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < marker_k - 1; i++)                        // :63-81
        for (int l = 0; l < 4; l++) {
            uint64_t nf = LUT.t[str[l][i]], nr = 3 - nf;
            f[l] = (f[l] << 2) | nf; r[l] = (r[l] >> 2) | (nr << 40);
        }
    uint64_t resume[4] = {0, 0, 0, 0};
    for (uint64_t i = marker_k - 1; i < len + marker_k - 1; i++) {     // :108
        for (int l = 0; l < 4; l++) {
            uint8_t b = str[l][i];
            if (b == 78) resume[l] = i + marker_k;                     // :115-126
            uint64_t nf = LUT.t[b], nr = 3 - nf;
            f[l] = ((f[l] << 2) | nf) & marker_mask;                   // :137-139
            r[l] = ((r[l] >> 2) & rev_marker_mask) | (nr << 40);       // :140-143
        }
        for (int l = 0; l < 4; l++) {                                  // lanes are emitted in order 0..3 (:181-270)
            uint64_t fs = f[l] & seed_mask, rs = r[l] & seed_mask;
            bool canon = rs > fs;                                      // :147
            uint64_t seed = canon ? fs : rs;                           // :149-150
            uint64_t h = mm_hash64(seed);
            if (h < threshold && resume[l] <= i) {
                sk.add_seed_position((uint32_t)seed, SeedPos{(uint32_t)(i + len * l), (contig << 1) | (canon ? 1u : 0u)});
                if (h < threshold_marker) sk.markers.insert(r[l] > f[l] ? f[l] : r[l]);  // :148,189-199
            }
        }
    }
}

#ifdef __AVX2__
// The same function with the reference's own instruction set (avx2_seeding.rs:33-272 IS AVX2 intrinsics code: four 64-bit lanes, one substring each).
// Results are identical to fmh_seeds_avx2_semantics above, which stays as its checker (tests/test_oracle_golden.py) and is what non-AVX2 builds run;
// this one is what sketching calls, so that the CPU baseline of bench.py is timed on vector code like the reference's.
inline __m256i mm_hash64_x4(__m256i key) {   // types.rs:86-96, four keys
    key = _mm256_xor_si256(_mm256_add_epi64(key, _mm256_slli_epi64(key, 21)), _mm256_set1_epi64x(-1));
    key = _mm256_xor_si256(key, _mm256_srli_epi64(key, 24));
    key = _mm256_add_epi64(_mm256_add_epi64(key, _mm256_slli_epi64(key, 3)), _mm256_slli_epi64(key, 8));
    key = _mm256_xor_si256(key, _mm256_srli_epi64(key, 14));
    key = _mm256_add_epi64(_mm256_add_epi64(key, _mm256_slli_epi64(key, 2)), _mm256_slli_epi64(key, 4));
    key = _mm256_xor_si256(key, _mm256_srli_epi64(key, 28));
    key = _mm256_add_epi64(key, _mm256_slli_epi64(key, 31));
    return key;
}
void fmh_seeds_avx2(const uint8_t* s, uint64_t slen, uint32_t c, uint32_t k, uint32_t marker_c, uint32_t contig, ora_sketch& sk) {
    const uint32_t marker_k = K_MARKER_DNA;
    const uint64_t len = (slen - marker_k + 1) / 4;                    // :48
    if (slen < 2 * marker_k) return;                                   // :56
    const __m256i seed_mask = _mm256_set1_epi64x((long long)(~0ull >> (64 - 2 * k)));
    const __m256i marker_mask = _mm256_set1_epi64x((long long)(~0ull >> (64 - 2 * marker_k)));
    const __m256i rev_marker_mask = _mm256_set1_epi64x((long long)~(3ull << (2 * marker_k - 2)));
    const uint64_t threshold = ~0ull / (uint64_t)c, threshold_marker = ~0ull / (uint64_t)marker_c;
    const __m256i sign = _mm256_set1_epi64x((long long)0x8000000000000000ull);
    const __m256i thr_s = _mm256_xor_si256(_mm256_set1_epi64x((long long)threshold), sign);     // unsigned compare through the signed one
    const __m256i three = _mm256_set1_epi64x(3);
    const uint8_t* str[4] = {s, s + len, s + 2 * len, s + 3 * len};    // :49-52
    __m256i f = _mm256_setzero_si256(), r = _mm256_setzero_si256();
    auto load = [&](uint64_t i) { return _mm256_set_epi64x(LUT.t[str[3][i]], LUT.t[str[2][i]], LUT.t[str[1][i]], LUT.t[str[0][i]]); };
    for (uint32_t i = 0; i < marker_k - 1; i++) {                      // :63-81
        const __m256i nf = load(i), nr = _mm256_sub_epi64(three, nf);
        f = _mm256_or_si256(_mm256_slli_epi64(f, 2), nf); r = _mm256_or_si256(_mm256_srli_epi64(r, 2), _mm256_slli_epi64(nr, 40));
    }
    uint64_t resume[4] = {0, 0, 0, 0};
    alignas(32) uint64_t fa[4], ra[4], sa[4], ha[4], ca[4];
    for (uint64_t i = marker_k - 1; i < len + marker_k - 1; i++) {     // :108
        for (int l = 0; l < 4; l++) if (str[l][i] == 78) resume[l] = i + marker_k;   // :115-126
        const __m256i nf = load(i), nr = _mm256_sub_epi64(three, nf);
        f = _mm256_and_si256(_mm256_or_si256(_mm256_slli_epi64(f, 2), nf), marker_mask);                                      // :137-139
        r = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi64(r, 2), rev_marker_mask), _mm256_slli_epi64(nr, 40));           // :140-143
        const __m256i fs = _mm256_and_si256(f, seed_mask), rs = _mm256_and_si256(r, seed_mask);
        const __m256i canon = _mm256_cmpgt_epi64(rs, fs);                                                                     // :147 (values below 2^32)
        const __m256i seed = _mm256_blendv_epi8(rs, fs, canon);                                                               // :149-150
        const __m256i h = mm_hash64_x4(seed);
        const int hit = _mm256_movemask_pd(_mm256_castsi256_pd(_mm256_cmpgt_epi64(thr_s, _mm256_xor_si256(h, sign))));        // h < threshold
        if (!hit) continue;
        _mm256_store_si256((__m256i*)fa, f); _mm256_store_si256((__m256i*)ra, r); _mm256_store_si256((__m256i*)sa, seed);
        _mm256_store_si256((__m256i*)ha, h); _mm256_store_si256((__m256i*)ca, canon);
        for (int l = 0; l < 4; l++) {                                  // lanes are emitted in order 0..3 (:181-270)
            if (!((hit >> l) & 1) || resume[l] > i) continue;
            sk.add_seed_position((uint32_t)sa[l], SeedPos{(uint32_t)(i + len * l), (contig << 1) | (ca[l] ? 1u : 0u)});
            if (ha[l] < threshold_marker) sk.markers.insert(ra[l] > fa[l] ? fa[l] : ra[l]);  // :148,189-199
        }
    }
}
#else
inline void fmh_seeds_avx2(const uint8_t* s, uint64_t slen, uint32_t c, uint32_t k, uint32_t marker_c, uint32_t contig, ora_sketch& sk) {
    fmh_seeds_avx2_semantics(s, slen, c, k, marker_c, contig, sk);
}
#endif

// ---------------------------------------------------------------- chaining data (types.rs:499-550)
struct Anchor { uint32_t qctg, qpos, rctg, rpos; bool rev; };
inline bool anchor_less(const Anchor& a, const Anchor& b) {  // derived Ord, types.rs:499-506
    if (a.qctg != b.qctg) return a.qctg < b.qctg;
    if (a.qpos != b.qpos) return a.qpos < b.qpos;
    if (a.rctg != b.rctg) return a.rctg < b.rctg;
    if (a.rpos != b.rpos) return a.rpos < b.rpos;
    return a.rev < b.rev;
}
struct ChainInterval {  // types.rs:508-519 (field order = comparison order)
    double score; size_t num_anchors; uint32_t q0, q1, r0, r1; size_t rctg, qctg, chunk_id; bool rev; uint32_t overlap;
};
inline int ci_cmp(const ChainInterval& a, const ChainInterval& b) {  // derived PartialOrd
#define CMPF(x) if (a.x < b.x) return -1; if (a.x > b.x) return 1;
    CMPF(score) CMPF(num_anchors) CMPF(q0) CMPF(q1) CMPF(r0) CMPF(r1) CMPF(rctg) CMPF(qctg) CMPF(chunk_id) CMPF(rev) CMPF(overlap)
#undef CMPF
    return 0;
}
// Scratch of chain_seeds.  A thread keeps ONE across its calls (thread_local in chain_seeds) and every stage works on flat arrays in it: the reference
// leans on its allocator (main.rs:10-18) for the per-pair vectors of chain.rs; with glibc's malloc each pair's half-megabyte anchor vector is an
// mmap / munmap pair, and many threads then queue on the process's address-space lock instead of chaining (measured: ~5 % parallel efficiency at
// 256 threads before this).  Storage only -- every stage below computes exactly what its reference lines compute, in the same order.
struct Work {
    std::vector<Anchor> anchors;                               // all anchors, sorted (chain.rs:721)
    std::vector<std::vector<uint32_t>> qpos_all;               // query_positions_all per query contig (the first n_qctg entries are live)
    std::vector<uint32_t> chunk_begin;                         // chunk i = anchors [chunk_begin[i], chunk_begin[i + 1])
    std::vector<uint32_t> seeds, seeds_begin;                  // seeds_in_chunk, flattened the same way
    std::vector<size_t> ptr, root, members, best;              // per anchor of the current chunk
    std::vector<double> score, max_score;
    std::vector<ChainInterval> ints;                           // candidate intervals of all chunks (get_chain_intervals)
    std::vector<ChainInterval> good; std::vector<uint32_t> good_begin, good_fill;   // accepted intervals grouped by chunk, acceptance order inside a chunk
    std::vector<uint32_t> accepted;                            // indices into ints, acceptance order
    std::vector<std::vector<uint32_t>> acc_q, acc_r;           // accepted intervals per query / ref contig (the reference's interval trees)
    std::vector<std::pair<double, size_t>> ests;
    std::vector<std::pair<uint32_t, uint32_t>> uni;
    std::vector<double> nomult, mult, boot; std::vector<size_t> rv;
    size_t n_chunks() const { return chunk_begin.empty() ? 0 : chunk_begin.size() - 1; }
};

double mean_len(const std::vector<uint32_t>& v) { double s = 0; for (auto x : v) s += (double)x; return s / (double)v.size(); }

// chain.rs:15-26
bool switch_qr(double mean_r, double mean_q, double q_sk_len, double r_sk_len, const std::string& qname, const std::string& rname) {
    double sq = q_sk_len * std::min(mean_q, 300000.), sr = r_sk_len * std::min(mean_r, 300000.);
    if (sq == sr) return qname > rname;
    return sq > sr;
}

// chain.rs:608-836
bool get_anchors(const ora_sketch& ref, const ora_sketch& query, size_t band, Work& w, bool& have, ora_chain_stats* st) {
    have = false;
    w.anchors.clear(); w.chunk_begin.clear(); w.seeds.clear(); w.seeds_begin.clear();
    if (ref.contig_lengths.empty() || query.contig_lengths.empty()) return true;  // :618-620
    double mean_q = mean_len(query.contig_lengths), mean_r = mean_len(ref.contig_lengths);
    double qproxy, rproxy;
    if (query.total_len > 100000 && ref.total_len > 100000) {                      // :641-648
        qproxy = (double)query.markers.size() * (double)query.c; rproxy = (double)ref.markers.size() * (double)ref.c;
    } else { qproxy = (double)query.total_len; rproxy = (double)ref.total_len; }
    bool switched = switch_qr(mean_r, mean_q, qproxy, rproxy, query.file_name, ref.file_name);
    const ora_sketch& A = switched ? ref : query;   // enumerated ("kmer_seeds_query")
    const ora_sketch& B = switched ? query : ref;   // probed ("kmer_seeds_ref")
    const size_t n_qctg = A.contig_lengths.size();
    if (w.qpos_all.size() < n_qctg) w.qpos_all.resize(n_qctg);
    for (size_t c = 0; c < n_qctg; c++) w.qpos_all[c].clear();
    auto& qpos_all = w.qpos_all; auto& anchors = w.anchors;
    for (size_t h = 0; h < A.seeds.keys.size(); h++) {                              // :666
        uint32_t seed = A.seeds.keys[h];
        if (seed == SeedMap::EMPTY) continue;
        const SeedPos* qp; SeedPos t1; size_t nq = A.get_at(h, &qp, &t1);
        if (nq > band) continue;                                                    // :674-676
        const SeedPos* rp; SeedPos t2; size_t nr = B.get(seed, &rp, &t2);
        if (nr == 0) { for (size_t i = 0; i < nq; i++) qpos_all[qp[i].cc >> 1].push_back(qp[i].pos); continue; }  // :682-685
        if (nr > band) continue;                                                    // :694-696
        for (size_t i = 0; i < nq; i++) qpos_all[qp[i].cc >> 1].push_back(qp[i].pos);
        for (size_t i = 0; i < nq; i++) for (size_t j = 0; j < nr; j++)
            anchors.push_back(Anchor{qp[i].cc >> 1, qp[i].pos, rp[j].cc >> 1, rp[j].pos, (rp[j].cc & 1) != (qp[i].cc & 1)});  // :703-711
    }
    if (anchors.empty()) return true;                                               // :714-720 (returns switched=true)
    std::sort(anchors.begin(), anchors.end(), anchor_less);                         // :721
    for (size_t c = 0; c < n_qctg; c++) std::sort(qpos_all[c].begin(), qpos_all[c].end());   // :722-724
    if (st) {
        st->n_anchors = anchors.size(); st->n_qpos = 0; for (size_t c = 0; c < n_qctg; c++) st->n_qpos += qpos_all[c].size();
        uint64_t hsh = 1469598103934665603ull;
        auto mix = [&](uint64_t x) { hsh ^= x; hsh *= 1099511628211ull; };
        for (auto& a : anchors) { mix(a.qctg); mix(a.qpos); mix(a.rctg); mix(a.rpos); mix(a.rev); }
        st->anchor_checksum = hsh;
    }
    const uint32_t FRAG = CHUNK_SIZE_DNA;
    uint32_t last = anchors[0].qctg; uint32_t end = anchors[0].qpos + FRAG; size_t rc = 0;  // :742-745
    w.chunk_begin.push_back(0); w.seeds_begin.push_back(0);
    for (size_t ai = 0; ai < anchors.size(); ai++) {
        const Anchor& a = anchors[ai];
        if (last != a.qctg || a.qpos > end) {                                       // :747 -- the current chunk (never empty here) closes in front of anchor ai
            const auto& v = qpos_all[last];
            while (rc < v.size() && v[rc] <= end) { w.seeds.push_back(v[rc]); rc++; }   // :755-780
            w.seeds_begin.push_back((uint32_t)w.seeds.size());
            end += FRAG;                                                            // :782
            w.chunk_begin.push_back((uint32_t)ai);
            if (last != a.qctg) { end = a.qpos + FRAG; rc = 0; }                    // :786-789
        }
        last = a.qctg;
    }
    {                                                                               // :794-824 (the open chunk holds at least the last anchor)
        const auto& v = qpos_all[last];
        while (rc < v.size() && v[rc] <= anchors.back().qpos) { w.seeds.push_back(v[rc]); rc++; }
        w.seeds_begin.push_back((uint32_t)w.seeds.size()); w.chunk_begin.push_back((uint32_t)anchors.size());
    }
    have = true;
    return switched;
}

// chain.rs:558-603 -- returns false when the reference returns f64::MIN
inline bool score_anchors(const Anchor& cur, const Anchor& past, double& out) {
    if (cur.rev != past.rev) return false;
    if (cur.rpos == past.rpos || cur.qpos == past.qpos) return false;
    double dq = std::fabs((double)cur.qpos - (double)past.qpos);
    double dr = cur.rev ? (double)past.rpos - (double)cur.rpos : (double)cur.rpos - (double)past.rpos;
    if (dq > D_MAX_LIN_LENGTH || dr > D_MAX_LIN_LENGTH) return false;
    if (dr <= 0.) return false;
    double gap = std::fabs(dr - dq);
    if (gap > D_MAX_GAP_LENGTH) return false;
    out = D_ANCHOR_SCORE_ANI - gap;
    return true;
}

// chain.rs:838-896 on one chunk (n anchors at ch): fills w.score / w.ptr
void chain_chunk(const Anchor* ch, size_t n, size_t band, Work& w) {
    const uint32_t past_len = (uint32_t)std::min<size_t>(CHUNK_SIZE_DNA / 2, BP_CHAIN_BAND);  // :842
    w.ptr.assign(n, 0); w.score.assign(n, 0.);
    for (size_t i = 0; i < n; i++) {
        double best = 0.; size_t bp = i;
        for (size_t j = i; j-- > 0;) {                                 // (0..i).rev()
            if (ch[i].rctg != ch[j].rctg) continue;                    // :856-858
            if (ch[i].qpos - ch[j].qpos > past_len || i - j > band) break;  // :859-863
            double s; if (!score_anchors(ch[i], ch[j], s)) continue;
            double ns = s + w.score[j];
            if (ns > best) { best = ns; bp = j; }                      // :876-879
        }
        w.score[i] = best; w.ptr[i] = bp;
    }
}

// chain.rs:939-1007.  Set membership = anchors sharing the pointer-forest root (chain.rs:883-885 unions
// i with ptr[i] only).  Set iteration order per partitions 0.2.4: root first, then
// members in descending index -- so the strict `>` of :952-964 keeps, among equal scores, the root, else the largest index.
void get_chain_intervals(Work& w, const Anchor* an, size_t n, size_t chunk_id) {
    w.root.resize(n); w.members.assign(n, 0); w.best.resize(n); w.max_score.resize(n);
    for (size_t i = 0; i < n; i++) {
        w.root[i] = w.ptr[i] == i ? i : w.root[w.ptr[i]];
        if (w.root[i] == i) { w.best[i] = i; w.max_score[i] = w.score[i]; }         // the root is visited first
    }
    for (size_t i = n; i-- > 0;) {                                                  // then the other members, descending
        const size_t r = w.root[i];
        if (r == i) continue;
        w.members[r]++;
        if (w.score[i] > w.max_score[r]) { w.max_score[r] = w.score[i]; w.best[r] = i; }
    }
    const double min_score = (double)D_MIN_ANCHORS_ANI * D_ANCHOR_SCORE_ANI * 0.75;  // chain.rs:113
    for (size_t r = 0; r < n; r++) {
        if (w.root[r] != r) continue;
        if (w.members[r] + 1 < D_MIN_ANCHORS_ANI) continue;                         // :954-957
        const double max_score = w.max_score[r]; const size_t best = w.best[r];     // :952-964
        size_t idx = best, na = 1;
        while (w.ptr[idx] != idx) { idx = w.ptr[idx]; na++; }                       // :969-973
        if (na < D_MIN_ANCHORS_ANI || max_score < min_score) continue;              // :974-977
        ChainInterval ci;
        ci.q0 = an[idx].qpos; ci.q1 = an[best].qpos;
        ci.r0 = std::min(an[idx].rpos, an[best].rpos); ci.r1 = std::max(an[idx].rpos, an[best].rpos);
        ci.rctg = an[idx].rctg; ci.qctg = an[idx].qctg; ci.score = max_score; ci.num_anchors = na;
        ci.chunk_id = chunk_id; ci.rev = an[idx].rev; ci.overlap = 0;
        w.ints.push_back(ci);
    }
}

// chain.rs:1008-1099 (bio IntervalTree::find(a..b) = stored s..e with s<b && a<e; only sums are used).  Output: w.good grouped by chunk.
void get_nonoverlapping(Work& w, size_t n_chunks, size_t n_qctg, size_t n_rctg, ora_chain_stats* st) {
    auto& ints = w.ints;
    std::stable_sort(ints.begin(), ints.end(), [](const ChainInterval& x, const ChainInterval& y) { return ci_cmp(y, x) < 0; });  // :1012
    if (w.acc_q.size() < n_qctg) w.acc_q.resize(n_qctg);
    if (w.acc_r.size() < n_rctg) w.acc_r.resize(n_rctg);
    for (const auto& in : ints) { w.acc_q[in.qctg].clear(); w.acc_r[in.rctg].clear(); }
    w.accepted.clear();
    uint64_t hsh = 1469598103934665603ull; uint64_t nacc = 0;
    auto mix = [&](uint64_t x) { hsh ^= x; hsh *= 1099511628211ull; };
    for (size_t i = 0; i < ints.size(); i++) {
        const ChainInterval& in = ints[i];
        auto& tr = w.acc_r[in.rctg]; auto& tq = w.acc_q[in.qctg];
        bool ok_ref, ok_q;
        {
            uint32_t sum = 0; size_t cnt = 0;
            for (uint32_t o : tr) { const ChainInterval& ol = ints[o]; if (ol.r0 < in.r1 && in.r0 < ol.r1) { cnt++; sum += std::min(in.r1 - ol.r0, ol.r1 - in.r0); } }
            ok_ref = cnt == 0 || (float)sum < (float)(in.r1 - in.r0) * OVERLAP_ORTHOLOGOUS_FRACTION;   // :1030-1056
        }
        {
            uint32_t sum = 0; size_t cnt = 0;
            for (uint32_t o : tq) { const ChainInterval& ol = ints[o]; if (ol.q0 < in.q1 && in.q0 < ol.q1) { cnt++; sum += std::min(in.q1 - ol.q0, ol.q1 - in.q0); } }
            ok_q = cnt == 0 || (float)sum < (float)(in.q1 - in.q0) * OVERLAP_ORTHOLOGOUS_FRACTION;     // :1058-1085
        }
        if (ok_ref && ok_q) {                                                        // :1086-1094 (stored overlap stays 0)
            tq.push_back((uint32_t)i); tr.push_back((uint32_t)i); w.accepted.push_back((uint32_t)i); nacc++;
            mix((uint64_t)in.score); mix(in.num_anchors); mix(in.q0); mix(in.q1); mix(in.r0); mix(in.r1); mix(in.rctg); mix(in.qctg); mix(in.chunk_id); mix(in.rev);
        }
    }
    // good_intervals[chunk_id].push(interval) in acceptance order (:1094): a stable bucketing by chunk
    w.good_begin.assign(n_chunks + 1, 0);
    for (uint32_t i : w.accepted) w.good_begin[ints[i].chunk_id + 1]++;
    for (size_t c = 0; c < n_chunks; c++) w.good_begin[c + 1] += w.good_begin[c];
    w.good_fill.assign(w.good_begin.begin(), w.good_begin.end() - 1);
    w.good.resize(w.accepted.size());
    for (uint32_t i : w.accepted) w.good[w.good_fill[ints[i].chunk_id]++] = ints[i];
    if (st) { st->n_accepted = nacc; st->interval_checksum = hsh; }
}

// fastrand 1.9.0 WyRand; used only by chain.rs:57-86
struct WyRand {
    uint64_t s;
    uint64_t gen() { s += 0xA0761D6478BD642Full; unsigned __int128 t = (unsigned __int128)s * (unsigned __int128)(s ^ 0xE7037ED1A0B428DBull); return (uint64_t)t ^ (uint64_t)(t >> 64); }
    uint64_t below(uint64_t n) {
        uint64_t r = gen(); unsigned __int128 m = (unsigned __int128)r * n; uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo < n) { uint64_t t = (0 - n) % n; while (lo < t) { r = gen(); m = (unsigned __int128)r * n; hi = (uint64_t)(m >> 64); lo = (uint64_t)m; } }
        return hi;
    }
};

double std_deviation(const std::vector<double>& d) {  // chain.rs:39-55
    if (d.empty()) return 0.;
    double s = 0; for (double x : d) s += x; double m = s / (double)d.size();
    double v = 0; for (double x : d) { double df = m - x; v += df * df; }
    return std::sqrt(v / (double)d.size());
}

void bootstrap_interval(Work& w, const std::vector<std::pair<double, size_t>>& est, double& lo, double& hi, double& sd) {  // chain.rs:57-86
    auto& nomult = w.nomult; nomult.clear(); for (auto& e : est) nomult.push_back(e.first);
    sd = std_deviation(nomult);
    size_t num_samp = est.size();
    if (num_samp < 10) { lo = 0.; hi = 1.; return; }
    auto& mult = w.mult; mult.clear(); for (auto& e : est) for (size_t i = 0; i < e.second; i++) mult.push_back(e.first);
    WyRand rng{7};
    const size_t iters = 100; auto& res = w.boot; res.clear();
    auto& rv = w.rv; rv.resize(num_samp);
    for (size_t it = 0; it < iters; it++) {
        for (size_t j = 0; j < num_samp; j++) rv[j] = (size_t)rng.below(mult.size());
        double s = 0; for (size_t j = 0; j < num_samp; j++) s += mult[rv[j]];
        res.push_back(s / (double)num_samp);
    }
    std::sort(res.begin(), res.end());
    lo = res[iters * 5 / 100 - 1]; hi = res[iters * 95 / 100 - 1];
}

float f32nan() { return std::numeric_limits<float>::quiet_NaN(); }

// chain.rs:173-555
void calculate_ani(Work& w, const ora_sketch& ref, const ora_sketch& query, const ora_map_opts& mo, bool switched, ora_ani_result& out, ora_chain_stats* st) {
    const uint32_t k = ref.k; const uint32_t c = ref.c;                // map_params.k = ref.k (chain.rs:115)
    const bool sensitive_af = c < 200;                                 // :184-190
    auto& ests = w.ests; ests.clear();
    uint32_t total_query_bases = 0, total_ref_range = 0, avg_chain_int_len = 0, num_chains = 0;
    const size_t n_chunks = w.n_chunks();
    for (size_t i = 0; i < n_chunks; i++) {
        const ChainInterval* ints = w.good.data() + w.good_begin[i]; const size_t n_ints = w.good_begin[i + 1] - w.good_begin[i];
        size_t total_anchors = 0; uint32_t tbcq = 0;
        uint32_t rq0 = UINT32_MAX, rq1 = 0;
        auto& uni = w.uni; uni.clear();
        for (size_t x = 0; x < n_ints; x++) {
            const ChainInterval& in = ints[x];
            total_anchors += in.num_anchors;
            if (in.q0 < rq0) rq0 = in.q0;
            if (in.q1 > rq1) rq1 = in.q1;
            if (!switched) tbcq += in.q1 - in.q0 + k + 2 * c; else tbcq += in.r1 - in.r0 + k + 2 * c;   // :223-237
            uint32_t start = (uint32_t)std::max((int32_t)in.q0 - (int32_t)c, 0), stop = in.q1 + c;      // :239-242
            uni.push_back({start, stop});
            if (sensitive_af) { total_query_bases += (in.q1 - in.q0) - in.overlap + 2 * c + k; total_ref_range += (in.q1 - in.q0) - in.overlap + 2 * c + k; }
            avg_chain_int_len += (in.q1 - in.q0) - in.overlap + 2 * c + k; num_chains += 1;             // :249-250
        }
        if (total_anchors == 0) continue;                                                                // :253
        if (rq1 - rq0 < MIN_LENGTH_COVER) continue;                                                      // :257
        if (!sensitive_af) { total_query_bases += rq1 - rq0 + 2 * c + k; total_ref_range += rq1 - rq0 + 2 * c + k; }
        size_t in_int = 0, upper_lower = 0;
        const uint32_t* sd0 = w.seeds.data() + w.seeds_begin[i]; const size_t n_seeds = w.seeds_begin[i + 1] - w.seeds_begin[i];
        for (size_t x = 0; x < n_seeds; x++) {
            const uint32_t p = sd0[x];
            bool hit = false; for (auto& u : uni) if (p >= u.first && p <= u.second) { hit = true; break; }
            if (hit) in_int++;
            if (p >= rq0 && p <= rq1) upper_lower++;                                                     // :326-332 (spacing = 0)
        }
        size_t considered = n_seeds;
        double putative = std::pow((double)total_anchors / (double)in_int, 1. / (double)k);              // :335-339
        if (putative > 0.950 && tbcq > c * 4 && rq1 - rq0 < (uint32_t)(CHUNK_SIZE_DNA * 9 / 10) &&
            (double)considered > 1.05 * (double)upper_lower)
            considered = upper_lower;                                                                    // :340-351
        double ml = std::min(1., (double)total_anchors / (double)considered);
        double ani_est = std::pow(ml, 1. / (double)k);                                                   // :377
        ests.push_back({ani_est, considered});
    }
    std::sort(ests.begin(), ests.end());                                                                 // :414
    if (st) st->n_estimates = ests.size();
    memset(&out, 0, sizeof out);
    if (ests.empty() || num_chains == 0) { out.ani = f32nan(); return; }                                 // :416-420 (AniEstResult::default())
    avg_chain_int_len /= num_chains;
    size_t total_mult = 0; for (auto& e : ests) total_mult += e.second;
    double lower = 0., upper = 1.;
    if (mo.median) { lower = 0.499; upper = 0.501; } else if (mo.robust) { lower = 0.10; upper = 0.90; }
    size_t lower_i = 0, upper_i = ests.size() - 1; bool cl = false; size_t cs = 0;
    for (size_t i = 0; i < ests.size(); i++) {                                                           // :449-460
        cs += ests[i].second;
        if (cs >= (size_t)((double)total_mult * lower) && !cl) { lower_i = i; cl = true; }
        if (cs >= (size_t)((double)total_mult * upper)) { upper_i = i + 1; break; }
    }
    size_t tm = 0; double wavg = 0.;
    for (size_t i = lower_i; i < upper_i; i++) { wavg += ests[i].first * (double)ests[i].second; tm += ests[i].second; }
    double final_ani = wavg / (double)tm;
    double ci_lo, ci_hi, sd; bootstrap_interval(w, ests, ci_lo, ci_hi, sd);
    double cov_q = std::min(1., (double)total_query_bases / (double)query.total_len);
    double cov_r = std::min(1., (double)total_ref_range / (double)ref.total_len);
    double cutoff = mo.min_af < 0. ? 0.15 : mo.min_af;                                                   // chain.rs:100-107
    if (mo.both_min_af > 0.0) { if (cov_q < mo.both_min_af || cov_r < mo.both_min_af) final_ani = -1.; }
    else if (cov_q < cutoff && cov_r < cutoff) final_ani = -1.;                                          // :500-517
    std::vector<uint32_t> sq = query.contig_lengths, sr = ref.contig_lengths;
    std::sort(sq.begin(), sq.end()); std::sort(sr.begin(), sr.end());
    size_t nq = sq.size(), nr = sr.size();
    out.ani = (float)final_ani; out.af_query = (float)cov_q; out.af_ref = (float)cov_r;
    out.num_contigs_r = (uint32_t)nr; out.num_contigs_q = (uint32_t)nq;
    out.ci_upper = (float)ci_hi; out.ci_lower = (float)ci_lo;
    out.q10_q = (float)sq[nq * 10 / 100]; out.q50_q = (float)sq[nq * 50 / 100]; out.q90_q = (float)sq[nq * 90 / 100];
    out.q10_r = (float)sr[nr * 10 / 100]; out.q50_r = (float)sr[nr * 50 / 100]; out.q90_r = (float)sr[nr * 90 / 100];
    out.std = (float)sd; out.avg_chain_int_len = avg_chain_int_len; out.total_bases_covered = total_query_bases;
}

float model_predict(const ora_model& m, const float* x) {  // gbdt 0.1.1 predict for LAD: bias + sum shrinkage*tree(x), f32
    float p = m.bias;
    for (uint32_t t = 0; t < m.n_trees; t++) {
        const ora_model::Node* nd = &m.nodes[m.off[t]]; int32_t i = 0;
        while (nd[i].feat >= 0) i = x[nd[i].feat] < nd[i].thr ? nd[i].left : nd[i].right;
        p += m.shrinkage * nd[i].pred;
    }
    return p;
}

// regression.rs:30-64
void predict_from_ani_res(ora_ani_result& r, const ora_model& m) {
    if (r.ani > 0.9f && r.total_bases_covered > TOTAL_BASES_REGRESS_CUTOFF) {
        float x[5];
        if (r.q50_r > r.q50_q) { x[0] = r.ani * 100.f; x[1] = r.std; x[2] = r.q90_r; x[3] = r.q90_q; x[4] = (float)r.avg_chain_int_len; }
        else { x[0] = r.ani * 100.f; x[1] = r.std; x[2] = r.q90_q; x[3] = r.q90_r; x[4] = (float)r.avg_chain_int_len; }
        float pred = model_predict(m, x);
        if (pred < 100.f) {
            r.ci_upper = (r.ci_upper - r.ani) + pred / 100.f;
            r.ci_lower = (r.ci_lower - r.ani) + pred / 100.f;
            r.ani = pred / 100.f;
        }
    }
}

void chain_seeds(const ora_sketch& ref, const ora_sketch& query, const ora_map_opts& mo, const ora_model* model,
                 ora_ani_result& out, ora_chain_stats* st) {
    static thread_local Work w;                                         // this thread's scratch, kept across pairs
    if (st) memset(st, 0, sizeof *st);
    const size_t band = BP_CHAIN_BAND / ref.c;                          // chain.rs:112 index_chain_band
    bool have;
    bool switched = get_anchors(ref, query, band, w, have, st);
    const size_t n_chunks = w.n_chunks();
    if (st) { st->switched = switched; st->n_chunks = n_chunks; }
    w.ints.clear();
    for (size_t i = 0; i < n_chunks; i++) {
        const Anchor* ch = w.anchors.data() + w.chunk_begin[i]; const size_t n = w.chunk_begin[i + 1] - w.chunk_begin[i];
        chain_chunk(ch, n, band, w); get_chain_intervals(w, ch, n, i);
    }
    if (st) st->n_intervals = w.ints.size();
    const ora_sketch& A = switched ? ref : query; const ora_sketch& B = switched ? query : ref;   // interval contigs: query side = A, ref side = B
    get_nonoverlapping(w, n_chunks, A.contig_lengths.size(), B.contig_lengths.size(), st);
    calculate_ani(w, ref, query, mo, switched, out, st);
    if (model) predict_from_ani_res(out, *model);
}

double powi(double a, int b) {  // compiler-rt __powidf2 (what f64::powi lowers to)
    const bool recip = b < 0; double r = 1;
    while (true) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
    return recip ? 1 / r : r;
}

}  // namespace

// ================================================================= C interface
extern "C" {

uint64_t ora_mm_hash64(uint64_t key) { return mm_hash64(key); }
double ora_powi(double x, int n) { return powi(x, n); }

ora_sketch* ora_sketch_new(uint32_t c, uint32_t k, uint32_t marker_c, const char* file_name) {
    ora_sketch* s = new ora_sketch(); s->c = c; s->k = k; s->marker_c = marker_c; s->file_name = file_name ? file_name : ""; return s;
}
void ora_sketch_free(ora_sketch* s) { delete s; }

int ora_sketch_add_contig(ora_sketch* s, const uint8_t* seq, uint64_t len, int mode, uint64_t min_len) {
    if (len < min_len) return 0;                                        // file_io.rs:176
    uint32_t j = (uint32_t)s->contig_lengths.size();
    s->contig_lengths.push_back((uint32_t)len); s->total_len += len;   // file_io.rs:180-182
    if (mode == 1 && len >= K_MARKER_DNA) fmh_seeds_avx2(seq, len, s->c, s->k, s->marker_c, j, *s);
    else if (mode == 2 && len >= K_MARKER_DNA) fmh_seeds_avx2_semantics(seq, len, s->c, s->k, s->marker_c, j, *s);   // the plain-C++ statement of mode 1
    else if (mode == 0) fmh_seeds_scalar(seq, len, s->c, s->k, s->marker_c, j, *s);
    return 1;
}

// glibc hands every vector beyond 128 KB to mmap and returns it with munmap: with many threads sketching / chaining at once those calls queue on the
// process's address-space lock.  The reference links an allocator built for this (main.rs:10-18); here the thresholds are raised once so that the
// arenas keep and reuse the memory.
static void tune_malloc_once() {
    static std::once_flag once;
    std::call_once(once, [] { mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 64 << 20); });
}

// file_io.rs:141-252 for a batch of files: genome g = contigs [genome_contig_off[g], genome_contig_off[g + 1]); files are sketched in parallel like the
// reference's par_iter over files (file_io.rs:147), `threads` workers pulling genomes from a shared counter
void ora_sketch_batch(uint32_t n_genomes, const uint64_t* genome_contig_off, const uint8_t* const* seq, const uint64_t* len, uint32_t c, uint32_t k,
                      uint32_t marker_c, const char* const* names, int mode, uint64_t min_len, int threads, ora_sketch** out) {
    tune_malloc_once();
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const uint32_t g = next.fetch_add(1); if (g >= n_genomes) break;
            ora_sketch* s = ora_sketch_new(c, k, marker_c, names ? names[g] : "");
            for (uint64_t x = genome_contig_off[g]; x < genome_contig_off[g + 1]; x++) ora_sketch_add_contig(s, seq[x], len[x], mode, min_len);
            out[g] = s;
        }
    };
    std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(worker); for (auto& t : th) t.join();
}

// fastx_to_sketches (file_io.rs:141-252) from files on disk: every file is read by the thread that sketches it (file_io.rs:147 par_iter), records are
// split the way needletail does for FASTA (header line without '>', sequence = the following lines without their line ends, up to the next line that
// starts with '>'), contigs shorter than min_len are skipped (:176).  Plain FASTA only -- what bench.py's end-to-end leg writes.  out[i] = the sketch of
// paths[i], or NULL when the file cannot be read or holds no kept contig (:230: such files are dropped).
void ora_sketch_files(uint32_t n_files, const char* const* paths, uint32_t c, uint32_t k, uint32_t marker_c, int mode, uint64_t min_len, int threads, ora_sketch** out) {
    tune_malloc_once();
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
        std::vector<char> data; std::vector<uint8_t> seq;
        for (;;) {
            const uint32_t f = next.fetch_add(1); if (f >= n_files) break;
            out[f] = nullptr;
            FILE* fp = fopen(paths[f], "rb"); if (!fp) continue;
            fseek(fp, 0, SEEK_END); const long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
            data.resize((size_t)(sz > 0 ? sz : 0));
            const size_t n = sz > 0 ? fread(data.data(), 1, (size_t)sz, fp) : 0;
            fclose(fp);
            ora_sketch* s = ora_sketch_new(c, k, marker_c, paths[f]);
            size_t p = 0; int kept = 0;
            while (p < n && data[p] != '>') p++;
            while (p < n) {
                const char* eol = (const char*)memchr(data.data() + p, '\n', n - p);
                p = eol ? (size_t)(eol - data.data()) + 1 : n;
                seq.clear();
                while (p < n && data[p] != '>') {
                    const char* le = (const char*)memchr(data.data() + p, '\n', n - p);
                    const size_t l = le ? (size_t)(le - data.data()) : n;
                    for (size_t i = p; i < l; i++) if (data[i] != '\r') seq.push_back((uint8_t)data[i]);
                    p = l < n ? l + 1 : n;
                }
                kept += ora_sketch_add_contig(s, seq.data(), seq.size(), mode, min_len);
            }
            if (kept) out[f] = s; else ora_sketch_free(s);
        }
    };
    std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(worker); for (auto& t : th) t.join();
}

ora_sketch* ora_sketch_from_arrays(uint32_t c, uint32_t k, uint32_t marker_c, const char* file_name, const uint32_t* seed,
                                   const uint32_t* pos, const uint32_t* ctgcanon, uint64_t n_pos, const uint64_t* markers,
                                   uint64_t n_markers, const uint32_t* contig_lengths, uint32_t n_contigs, uint64_t total_len) {
    ora_sketch* s = ora_sketch_new(c, k, marker_c, file_name);
    for (uint64_t i = 0; i < n_pos; i++) s->add_seed_position(seed[i], SeedPos{pos[i], ctgcanon[i]});
    for (uint64_t i = 0; i < n_markers; i++) s->markers.insert(markers[i]);
    s->contig_lengths.assign(contig_lengths, contig_lengths + n_contigs); s->total_len = total_len;
    return s;
}

uint64_t ora_sketch_n_positions(const ora_sketch* s) { return s->n_positions; }
uint64_t ora_sketch_n_distinct(const ora_sketch* s) { return s->seeds.n; }
uint64_t ora_sketch_n_markers(const ora_sketch* s) { return s->markers.size(); }
uint32_t ora_sketch_n_contigs(const ora_sketch* s) { return (uint32_t)s->contig_lengths.size(); }
uint64_t ora_sketch_total_len(const ora_sketch* s) { return s->total_len; }

static void collect(const ora_sketch* s, std::vector<std::array<uint32_t, 3>>& v) {
    for (size_t h = 0; h < s->seeds.keys.size(); h++) {
        uint32_t seed = s->seeds.keys[h]; if (seed == SeedMap::EMPTY) continue;
        const SeedPos* p; SeedPos t; size_t n = s->get(seed, &p, &t);
        for (size_t i = 0; i < n; i++) v.push_back({seed, p[i].pos, p[i].cc});
    }
}
void ora_sketch_export_seeds(const ora_sketch* s, uint32_t* seed, uint32_t* pos, uint32_t* cc) {
    std::vector<std::array<uint32_t, 3>> v; collect(s, v);
    std::sort(v.begin(), v.end(), [](const std::array<uint32_t, 3>& a, const std::array<uint32_t, 3>& b) {
        if (a[0] != b[0]) return a[0] < b[0];
        if ((a[2] >> 1) != (b[2] >> 1)) return (a[2] >> 1) < (b[2] >> 1);
        return a[1] < b[1]; });
    for (size_t i = 0; i < v.size(); i++) { seed[i] = v[i][0]; pos[i] = v[i][1]; cc[i] = v[i][2]; }
}
void ora_sketch_export_seeds_pos_order(const ora_sketch* s, uint32_t* seed, uint32_t* pos, uint32_t* cc) {
    std::vector<std::array<uint32_t, 3>> v; collect(s, v);
    std::sort(v.begin(), v.end(), [](const std::array<uint32_t, 3>& a, const std::array<uint32_t, 3>& b) {
        if ((a[2] >> 1) != (b[2] >> 1)) return (a[2] >> 1) < (b[2] >> 1);
        return a[1] < b[1]; });
    for (size_t i = 0; i < v.size(); i++) { seed[i] = v[i][0]; pos[i] = v[i][1]; cc[i] = v[i][2]; }
}
void ora_sketch_export_markers(const ora_sketch* s, uint64_t* out) {
    size_t i = 0; s->markers.for_each([&](uint64_t m) { out[i++] = m; }); std::sort(out, out + i);
}
void ora_sketch_export_contig_lengths(const ora_sketch* s, uint32_t* lens) { memcpy(lens, s->contig_lengths.data(), 4 * s->contig_lengths.size()); }

ora_model* ora_model_load(const char* path) {
    FILE* f = fopen(path, "rb"); if (!f) return nullptr;
    char magic[4]; uint32_t nt, nf, nn; float sh, bias;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "GBDT", 4) || fread(&nt, 4, 1, f) != 1 || fread(&nf, 4, 1, f) != 1 ||
        fread(&sh, 4, 1, f) != 1 || fread(&bias, 4, 1, f) != 1 || fread(&nn, 4, 1, f) != 1) { fclose(f); return nullptr; }
    ora_model* m = new ora_model(); m->n_trees = nt; m->n_feat = nf; m->shrinkage = sh; m->bias = bias;
    m->off.resize(nt + 1); m->nodes.resize(nn);
    bool ok = fread(m->off.data(), 4, nt + 1, f) == nt + 1 && fread(m->nodes.data(), sizeof(ora_model::Node), nn, f) == nn;
    fclose(f); if (!ok) { delete m; return nullptr; }
    return m;
}
void ora_model_free(ora_model* m) { delete m; }
float ora_model_predict(const ora_model* m, const float feat[5]) { return model_predict(*m, feat); }

void ora_chain_seeds(const ora_sketch* ref, const ora_sketch* query, const ora_map_opts* mo, const ora_model* model,
                     ora_ani_result* out, ora_chain_stats* stats) {
    chain_seeds(*ref, *query, *mo, model, *out, stats);
}

// screen.rs:84-142
int ora_check_markers_quickly(const ora_sketch* ref, const ora_sketch* query, double screen_val, int rescue_small) {
    if (screen_val == 0.) return 1;
    const MarkerSet *s1, *s2; size_t min_card;
    if (query->markers.size() > ref->markers.size()) { s1 = &ref->markers; s2 = &query->markers; min_card = ref->markers.size(); }
    else { s2 = &ref->markers; s1 = &query->markers; min_card = query->markers.size(); }
    if (min_card < SCREEN_MINIMUM_KMERS && rescue_small) return 1;
    if (min_card == 0) return rescue_small ? 1 : 0;
    size_t ratio = (size_t)(powi(screen_val, (int)K_MARKER_DNA) * (double)min_card);
    if (ratio == 0) ratio = 1;
    size_t inter = 0; bool ok = false;
    for (uint64_t m : s1->keys) {
        if (m == MarkerSet::EMPTY) continue;
        if (s2->contains(m)) inter++;
        if (inter >= ratio) { ok = true; break; }
    }
    return ok ? 1 : 0;
}

namespace {
struct InvIndex {  // screen.rs:190-210 (marker -> genome ids), built sort-based
    std::vector<uint64_t> key; std::vector<uint32_t> gid; std::vector<uint64_t> ukey; std::vector<uint32_t> ustart; size_t mask = 0; std::vector<uint32_t> table;
    // The reference builds this map serially (triangle.rs:55 -> screen.rs:190-210).  With threads > 1 the (marker, genome) incidences are dealt into
    // ranges of the marker value, every range is sorted by one thread and the ranges are concatenated: the same sorted list, the same table.
    void build(const ora_sketch* const* sk, uint32_t n, int threads = 1) {
        std::vector<std::pair<uint64_t, uint32_t>> v;
        if (threads <= 1 || n < 64) {
            for (uint32_t g = 0; g < n; g++) sk[g]->markers.for_each([&](uint64_t m) { v.push_back({m, g}); });
            std::sort(v.begin(), v.end());
        } else {
            const uint32_t NB = (uint32_t)threads * 4;                               // ranges of the 42-bit marker value
            auto bucket = [&](uint64_t m) { return (uint32_t)(((m >> 10) * NB) >> 32); };
            std::vector<std::vector<std::vector<std::pair<uint64_t, uint32_t>>>> part(threads, std::vector<std::vector<std::pair<uint64_t, uint32_t>>>(NB));
            std::atomic<uint32_t> next{0};
            auto deal = [&](int t) { for (;;) { const uint32_t g = next.fetch_add(1); if (g >= n) break; sk[g]->markers.for_each([&](uint64_t m) { part[t][bucket(m)].push_back({m, g}); }); } };
            { std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(deal, t); for (auto& t : th) t.join(); }
            std::vector<size_t> off(NB + 1, 0);
            for (uint32_t b = 0; b < NB; b++) { size_t c = 0; for (int t = 0; t < threads; t++) c += part[t][b].size(); off[b + 1] = off[b] + c; }
            v.resize(off[NB]);
            std::atomic<uint32_t> nb{0};
            auto sort_range = [&]() { for (;;) { const uint32_t b = nb.fetch_add(1); if (b >= NB) break; size_t at = off[b];
                for (int t = 0; t < threads; t++) { std::copy(part[t][b].begin(), part[t][b].end(), v.begin() + at); at += part[t][b].size(); }
                std::sort(v.begin() + off[b], v.begin() + off[b + 1]); } };
            { std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(sort_range); for (auto& t : th) t.join(); }
        }
        key.resize(v.size()); gid.resize(v.size());
        for (size_t i = 0; i < v.size(); i++) { key[i] = v[i].first; gid[i] = v[i].second; if (i == 0 || v[i].first != v[i - 1].first) { ukey.push_back(v[i].first); ustart.push_back((uint32_t)i); } }
        ustart.push_back((uint32_t)v.size());
        size_t c = 16; while (c < ukey.size() * 2) c <<= 1; mask = c - 1; table.assign(c, UINT32_MAX);
        for (uint32_t u = 0; u < ukey.size(); u++) { size_t h = mm_hash64(ukey[u]) & mask; while (table[h] != UINT32_MAX) h = (h + 1) & mask; table[h] = u; }
    }
    bool find(uint64_t m, uint32_t& s, uint32_t& e) const {
        size_t h = mm_hash64(m) & mask;
        while (table[h] != UINT32_MAX) { uint32_t u = table[h]; if (ukey[u] == m) { s = ustart[u]; e = ustart[u + 1]; return true; } h = (h + 1) & mask; }
        return false;
    }
};

// screen.rs:148-189 (rule 0) and :39-77 (rule 2)
void screen_row(const InvIndex& ix, const ora_sketch* const* refs, uint32_t n_refs, const ora_sketch* q, double identity, int rule,
                int rescue_small, std::vector<uint32_t>& cnt, std::vector<uint32_t>& touched, std::vector<uint32_t>& out) {
    out.clear();
    if (rule == 0 && q->markers.size() < 20 && rescue_small) { for (uint32_t i = 0; i < n_refs; i++) out.push_back(i); return; }
    touched.clear();
    q->markers.for_each([&](uint64_t m) {
        uint32_t s, e; if (!ix.find(m, s, e)) return;
        for (uint32_t i = s; i < e; i++) { uint32_t g = ix.gid[i]; if (cnt[g]++ == 0) touched.push_back(g); }
    });
    double cutoff = powi(identity, (int)K_MARKER_DNA);
    for (uint32_t g : touched) {
        size_t mn = std::min(refs[g]->markers.size(), q->markers.size());
        size_t thr = std::max((size_t)(cutoff * (double)mn), (size_t)1);
        if ((size_t)cnt[g] > thr) out.push_back(g);
        cnt[g] = 0;
    }
    std::sort(out.begin(), out.end());
}
}  // namespace

uint64_t ora_screen_refs(const ora_sketch* const* refs, uint32_t n_refs, const ora_sketch* query, double identity, int rule,
                         int rescue_small, uint32_t* out_ids) {
    InvIndex ix; ix.build(refs, n_refs);
    std::vector<uint32_t> cnt(n_refs, 0), touched, out;
    screen_row(ix, refs, n_refs, query, identity, rule, rescue_small, cnt, touched, out);
    memcpy(out_ids, out.data(), 4 * out.size());
    return out.size();
}

// wall-clock of the three phases of the last ora_triangle call (inverted index, screen, chain) -- bench.py's cpu_baseline reports them
static double g_tri_phase[3] = {0, 0, 0};
void ora_triangle_phases(double* index_s, double* screen_s, double* chain_s) { *index_s = g_tri_phase[0]; *screen_s = g_tri_phase[1]; *chain_s = g_tri_phase[2]; }

// triangle.rs:33-105
uint64_t ora_triangle(const ora_sketch* const* sk, uint32_t n, double screen_val, int rescue_small, const ora_map_opts* mo,
                      const ora_model* model, int threads, uint32_t* out_i, uint32_t* out_j, ora_ani_result* out_res, uint64_t cap,
                      uint64_t* n_chained, uint64_t* n_screen_pass) {
    if (screen_val == 0.) screen_val = 0.80;                            // triangle.rs:33-42
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    tune_malloc_once();
    const auto t_start = std::chrono::steady_clock::now();
    InvIndex ix; ix.build(sk, n, threads);                              // triangle.rs:55
    const auto t_index = std::chrono::steady_clock::now();
    std::vector<std::vector<uint32_t>> pass(n);
    std::atomic<uint32_t> next{0};
    auto screen_worker = [&]() {
        std::vector<uint32_t> cnt(n, 0), touched, out;
        for (;;) { uint32_t i = next.fetch_add(1); if (i + 1 >= n) break;   // rows 0..n-2 (triangle.rs:71)
            screen_row(ix, sk, n, sk[i], screen_val, 0, rescue_small, cnt, touched, out);
            for (uint32_t j : out) if (j > i) pass[i].push_back(j); }       // triangle.rs:90
    };
    { std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(screen_worker); for (auto& t : th) t.join(); }
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    for (uint32_t i = 0; i < n; i++) for (uint32_t j : pass[i]) pairs.push_back({i, j});
    if (n_screen_pass) *n_screen_pass = pairs.size();
    const auto t_screen = std::chrono::steady_clock::now();
    std::vector<ora_ani_result> res(pairs.size());
    std::atomic<uint64_t> nextp{0};
    auto chain_worker = [&]() {
        for (;;) { uint64_t p = nextp.fetch_add(1); if (p >= pairs.size()) break;
            chain_seeds(*sk[pairs[p].first], *sk[pairs[p].second], *mo, model, res[p], nullptr); }  // triangle.rs:98
    };
    { std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(chain_worker); for (auto& t : th) t.join(); }
    if (n_chained) *n_chained = pairs.size();
    const auto t_chain = std::chrono::steady_clock::now();
    g_tri_phase[0] = std::chrono::duration<double>(t_index - t_start).count(); g_tri_phase[1] = std::chrono::duration<double>(t_screen - t_index).count();
    g_tri_phase[2] = std::chrono::duration<double>(t_chain - t_screen).count();
    uint64_t kept = 0;
    for (size_t p = 0; p < pairs.size(); p++)
        if (res[p].ani > 0.1f) { if (kept < cap) { out_i[kept] = pairs[p].first; out_j[kept] = pairs[p].second; out_res[kept] = res[p]; } kept++; }  // triangle.rs:99
    return kept;
}

// search.rs:97-200 (all reference sketches resident)
uint64_t ora_search(const ora_sketch* const* refs, uint32_t n_refs, const ora_sketch* const* queries, uint32_t n_queries, double screen_val, int use_index,
                    const ora_map_opts* mo, const ora_model* model, int threads, uint32_t* out_q, uint32_t* out_r, ora_ani_result* out_res, uint64_t cap,
                    uint64_t* n_chained) {
    if (screen_val == 0.) screen_val = 0.80;                            // SEARCH_ANI_CUTOFF_DEFAULT (params.rs)
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    tune_malloc_once();
    const auto t_start = std::chrono::steady_clock::now();
    InvIndex ix; if (use_index) ix.build(refs, n_refs, threads);       // search.rs:60-66
    const auto t_index = std::chrono::steady_clock::now();
    std::vector<std::vector<uint32_t>> pass(n_queries);
    std::atomic<uint32_t> next{0};
    auto screen_worker = [&]() {
        std::vector<uint32_t> cnt(n_refs, 0), touched, out;
        for (;;) {
            const uint32_t q = next.fetch_add(1); if (q >= n_queries) break;
            if (use_index) { screen_row(ix, refs, n_refs, queries[q], screen_val, 2, 0, cnt, touched, out); pass[q] = out; std::sort(pass[q].begin(), pass[q].end()); }   // search.rs:133-140
            else for (uint32_t r = 0; r < n_refs; r++) if (ora_check_markers_quickly(queries[q], refs[r], screen_val, 0)) pass[q].push_back(r);   // search.rs:122-131 (argument order as there)
        }
    };
    { std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(screen_worker); for (auto& t : th) t.join(); }
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    for (uint32_t q = 0; q < n_queries; q++) for (uint32_t r : pass[q]) pairs.push_back({q, r});
    const auto t_screen = std::chrono::steady_clock::now();
    std::vector<ora_ani_result> res(pairs.size());
    std::atomic<uint64_t> nextp{0};
    auto chain_worker = [&]() {
        for (;;) { const uint64_t p = nextp.fetch_add(1); if (p >= pairs.size()) break;
            chain_seeds(*refs[pairs[p].second], *queries[pairs[p].first], *mo, model, res[p], nullptr); }   // search.rs:176
    };
    { std::vector<std::thread> th; for (int t = 0; t < threads; t++) th.emplace_back(chain_worker); for (auto& t : th) t.join(); }
    if (n_chained) *n_chained = pairs.size();
    const auto t_chain = std::chrono::steady_clock::now();
    g_tri_phase[0] = std::chrono::duration<double>(t_index - t_start).count(); g_tri_phase[1] = std::chrono::duration<double>(t_screen - t_index).count();
    g_tri_phase[2] = std::chrono::duration<double>(t_chain - t_screen).count();
    uint64_t kept = 0;
    for (size_t p = 0; p < pairs.size(); p++)
        if (res[p].ani > 0.5f) { if (kept < cap) { out_q[kept] = pairs[p].first; out_r[kept] = pairs[p].second; out_res[kept] = res[p]; } kept++; }   // search.rs:178
    return kept;
}

}  // extern "C"
