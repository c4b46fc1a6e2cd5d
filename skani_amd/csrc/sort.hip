// sort.hip -- stable LSD radix sorts over whole device arrays (rocPRIM).  Utility, not a hot kernel: the marker sets of genomes beyond the LDS sort, the work
// order of a chain batch of more than 131,072 pairs, and the screen's incidence list when a bucket of its own sort (screen_keys.hip) outgrows the LDS -- none
// of which happens in the headline step, whose kernel trace holds no rocPRIM kernel since round 5.
// (The test suite's CPU kernel simulator links tests/emu/emu_sort.cpp in place of this file.)
#include <cstring>  // rocprim's texture iterator needs memset declared first

#include "internal.h"

#include <rocprim/rocprim.hpp>

namespace skh {

void sort_pairs_u32_u32(skh_ctx* ctx, uint32_t*& keys, uint32_t*& vals, uint64_t n, int end_bit) {
    if (n < 2) return;
    // sorted output lands in fresh arena arrays; the caller's pointers are redirected instead of copying 8 B per record back
    uint32_t* keys_out = ctx->arena.get<uint32_t>(n);
    uint32_t* vals_out = ctx->arena.get<uint32_t>(n);
    size_t tmp_bytes = 0;
    hip_check(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_out, vals, vals_out, n, 0, end_bit, ctx->stream), "radix_sort_pairs size");
    void* tmp = ctx->arena.take(tmp_bytes);
    hip_check(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, vals, vals_out, n, 0, end_bit, ctx->stream), "radix_sort_pairs");
    keys = keys_out; vals = vals_out;
}

// keys -> out (another array of n entries; `keys` is scratch afterwards): no copy back
void sort_keys_u64_into(skh_ctx* ctx, uint64_t* keys, uint64_t* out, uint64_t n, int end_bit, DBuf<char>* own_tmp) {
    if (n == 0) return;
    if (n == 1) { d2d(out, keys, 8, ctx->stream); return; }
    size_t tmp_bytes = 0;
    hip_check(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, out, n, 0, end_bit, ctx->stream), "radix_sort_keys size");
    void* tmp;
    if (own_tmp) { own_tmp->alloc(tmp_bytes ? tmp_bytes : 16); tmp = own_tmp->p; } else tmp = ctx->arena.take(tmp_bytes ? tmp_bytes : 16);
    hip_check(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, out, n, 0, end_bit, ctx->stream), "radix_sort_keys");
}

void sort_keys_u64(skh_ctx* ctx, uint64_t* keys, uint64_t n, int end_bit, int begin_bit) {
    if (n < 2) return;
    if (begin_bit != 0) throw Error("sort_keys_u64: rocPRIM's radix sort returns unsorted output for mid-sized inputs when begin_bit > 0 (tools/exp/rocprim_bits.hip)");
    uint64_t* keys_out = ctx->arena.get<uint64_t>(n);
    size_t tmp_bytes = 0;
    hip_check(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, keys_out, n, begin_bit, end_bit, ctx->stream), "radix_sort_keys size");
    void* tmp = ctx->arena.take(tmp_bytes);
    hip_check(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, keys_out, n, begin_bit, end_bit, ctx->stream), "radix_sort_keys");
    d2d(keys, keys_out, n * sizeof(uint64_t), ctx->stream);
}

// keys of segment s = [off[s], off[s + 1]) sorted on bits [0, end_bit) within the segment (the marker sets of a batch: ~5,000 keys per genome --
// rocPRIM sorts segments of that size in one workgroup each, one launch instead of seven device-wide radix passes)
// returns where the sorted keys are (an arena array, or `keys` itself): no copy back
uint64_t* sort_segments_u64(skh_ctx* ctx, uint64_t* keys, uint64_t n, uint32_t n_seg, const uint64_t* d_off, const uint64_t* h_off, int end_bit) {
    if (n < 2 || !n_seg) return keys;
    (void)h_off;
    uint64_t* keys_out = ctx->arena.get<uint64_t>(n);
    size_t tmp_bytes = 0;
    hip_check(rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, keys, keys_out, (unsigned int)n, n_seg, d_off, d_off + 1, 0, end_bit, ctx->stream), "segmented_radix_sort_keys size");
    void* tmp = ctx->arena.take(tmp_bytes ? tmp_bytes : 16);
    hip_check(rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, keys, keys_out, (unsigned int)n, n_seg, d_off, d_off + 1, 0, end_bit, ctx->stream), "segmented_radix_sort_keys");
    return keys_out;
}

}  // namespace skh
