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

void sort_keys_u64(skh_ctx* ctx, uint64_t* keys, uint64_t n, int end_bit, int begin_bit) {
    if (n < 2) return;
#ifndef SKANI_EMU
    uint64_t* keys_out = ctx->arena.get<uint64_t>(n);
    size_t tmp_bytes = 0;
    hip_check(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys_out, n, begin_bit, end_bit, ctx->stream), "radix_sort_keys size");
    void* tmp = ctx->arena.take(tmp_bytes);
    hip_check(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, keys_out, n, begin_bit, end_bit, ctx->stream), "radix_sort_keys");
    d2d(keys, keys_out, n * sizeof(uint64_t), ctx->stream);
#else
    (void)ctx; (void)end_bit;
    const uint64_t mask = begin_bit ? ~((1ull << begin_bit) - 1ull) : ~0ull;         // stable on the selected bits, like the radix sort
    std::stable_sort(keys, keys + n, [&](uint64_t a, uint64_t b) { return (a & mask) < (b & mask); });
#endif
}

}  // namespace skh
