// sort.hip -- stable LSD radix sorts over whole device arrays (rocPRIM).  Utility, not a hot kernel: it runs
// once per sketch set (seed order, marker sets) and once per screen (inverted index).
#include <cstring>  // rocprim's texture iterator needs memset declared first

#include "internal.h"

#ifndef SKANI_EMU
#include <rocprim/rocprim.hpp>
#else
#include <algorithm>
#include <numeric>
#endif

namespace skh {

void sort_pairs_u32_u32(skh_ctx* ctx, uint32_t*& keys, uint32_t*& vals, uint64_t n, int end_bit) {
    if (n < 2) return;
#ifndef SKANI_EMU
    // sorted output lands in fresh arena arrays; the caller's pointers are redirected instead of copying 8 B per record back
    uint32_t* keys_out = ctx->arena.get<uint32_t>(n);
    uint32_t* vals_out = ctx->arena.get<uint32_t>(n);
    size_t tmp_bytes = 0;
    hip_check(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_out, vals, vals_out, n, 0, end_bit, ctx->stream), "radix_sort_pairs size");
    void* tmp = ctx->arena.take(tmp_bytes);
    hip_check(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, vals, vals_out, n, 0, end_bit, ctx->stream), "radix_sort_pairs");
    keys = keys_out; vals = vals_out;
#else
    (void)ctx; (void)end_bit;
    std::vector<uint64_t> idx(n); std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return keys[a] < keys[b]; });
    std::vector<uint32_t> k(n), v(n);
    for (uint64_t i = 0; i < n; i++) { k[i] = keys[idx[i]]; v[i] = vals[idx[i]]; }
    memcpy(keys, k.data(), n * 4); memcpy(vals, v.data(), n * 4);
#endif
}

// keys -> out (another array of n entries; `keys` is scratch afterwards): no copy back
void sort_keys_u64_into(skh_ctx* ctx, uint64_t* keys, uint64_t* out, uint64_t n, int end_bit) {
    if (n == 0) return;
#ifndef SKANI_EMU
    if (n == 1) { d2d(out, keys, 8, ctx->stream); return; }
    size_t tmp_bytes = 0;
    hip_check(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, out, n, 0, end_bit, ctx->stream), "radix_sort_keys size");
    void* tmp = ctx->arena.take(tmp_bytes ? tmp_bytes : 16);
    hip_check(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, out, n, 0, end_bit, ctx->stream), "radix_sort_keys");
#else
    (void)ctx;
    const uint64_t mask = end_bit < 64 ? (1ull << end_bit) - 1ull : ~0ull;
    memcpy(out, keys, n * 8);
    std::stable_sort(out, out + n, [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
#endif
}

void sort_keys_u64(skh_ctx* ctx, uint64_t* keys, uint64_t n, int end_bit, int begin_bit) {
    if (n < 2) return;
    if (begin_bit != 0) throw Error("sort_keys_u64: rocPRIM's radix sort returns unsorted output for mid-sized inputs when begin_bit > 0 (tools/exp/rocprim_bits.hip)");
#ifndef SKANI_EMU
    uint64_t* keys_out = ctx->arena.get<uint64_t>(n);
    size_t tmp_bytes = 0;
    hip_check(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys_out, n, begin_bit, end_bit, ctx->stream), "radix_sort_keys size");
    void* tmp = ctx->arena.take(tmp_bytes);
    hip_check(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, keys_out, n, begin_bit, end_bit, ctx->stream), "radix_sort_keys");
    d2d(keys, keys_out, n * sizeof(uint64_t), ctx->stream);
#else
    (void)ctx;
    uint64_t mask = begin_bit ? ~((1ull << begin_bit) - 1ull) : ~0ull;               // stable on the selected bits, like the radix sort
    if (end_bit < 64) mask &= (1ull << end_bit) - 1ull;
    std::stable_sort(keys, keys + n, [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
#endif
}

// keys of segment s = [off[s], off[s + 1]) sorted on bits [0, end_bit) within the segment (the marker sets of a batch: ~5,000 keys per genome --
// rocPRIM sorts segments of that size in one workgroup each, one launch instead of seven device-wide radix passes)
// returns where the sorted keys are (an arena array, or `keys` itself): no copy back
uint64_t* sort_segments_u64(skh_ctx* ctx, uint64_t* keys, uint64_t n, uint32_t n_seg, const uint64_t* d_off, const uint64_t* h_off, int end_bit) {
    if (n < 2 || !n_seg) return keys;
#ifndef SKANI_EMU
    (void)h_off;
    uint64_t* keys_out = ctx->arena.get<uint64_t>(n);
    size_t tmp_bytes = 0;
    hip_check(rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, keys, keys_out, (unsigned int)n, n_seg, d_off, d_off + 1, 0, end_bit, ctx->stream), "segmented_radix_sort_keys size");
    void* tmp = ctx->arena.take(tmp_bytes ? tmp_bytes : 16);
    hip_check(rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, keys, keys_out, (unsigned int)n, n_seg, d_off, d_off + 1, 0, end_bit, ctx->stream), "segmented_radix_sort_keys");
    return keys_out;
#else
    (void)ctx; (void)d_off;
    const uint64_t mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1ull);
    for (uint32_t sg = 0; sg < n_seg; sg++)
        std::stable_sort(keys + h_off[sg], keys + h_off[sg + 1], [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
    return keys;
#endif
}

}  // namespace skh
