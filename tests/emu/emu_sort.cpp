// emu_sort.cpp -- TEST-ONLY: the sorts of skani_amd/csrc/sort.hip (rocPRIM on the GPU) for the CPU kernel simulator build: std::stable_sort on
// the same bit ranges.  Compiled into tests/emu/libskani_emu.so in place of sort.hip; never part of libskani_hip.so.
#include <algorithm>
#include <numeric>
#include <vector>

#include "internal.h"

namespace skh {

void sort_pairs_u32_u32(skh_ctx*, uint32_t*& keys, uint32_t*& vals, uint64_t n, int) {
    if (n < 2) return;
    std::vector<uint64_t> idx(n); std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return keys[a] < keys[b]; });
    std::vector<uint32_t> k(n), v(n);
    for (uint64_t i = 0; i < n; i++) { k[i] = keys[idx[i]]; v[i] = vals[idx[i]]; }
    memcpy(keys, k.data(), n * 4); memcpy(vals, v.data(), n * 4);
}

void sort_keys_u64_into(skh_ctx*, uint64_t* keys, uint64_t* out, uint64_t n, int end_bit, DBuf<char>*) {
    if (n == 0) return;
    const uint64_t mask = end_bit < 64 ? (1ull << end_bit) - 1ull : ~0ull;
    memcpy(out, keys, n * 8);
    std::stable_sort(out, out + n, [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
}

void sort_keys_u64(skh_ctx*, uint64_t* keys, uint64_t n, int end_bit, int begin_bit) {
    if (n < 2) return;
    if (begin_bit != 0) throw Error("sort_keys_u64: begin_bit > 0 is not supported (see sort.hip)");
    const uint64_t mask = end_bit < 64 ? (1ull << end_bit) - 1ull : ~0ull;
    std::stable_sort(keys, keys + n, [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
}

uint64_t* sort_segments_u64(skh_ctx*, uint64_t* keys, uint64_t n, uint32_t n_seg, const uint64_t*, const uint64_t* h_off, int end_bit) {
    if (n < 2 || !n_seg) return keys;
    const uint64_t mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1ull);
    for (uint32_t sg = 0; sg < n_seg; sg++)
        std::stable_sort(keys + h_off[sg], keys + h_off[sg + 1], [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
    return keys;
}

}  // namespace skh
