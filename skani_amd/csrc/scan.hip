// scan.hip -- exclusive prefix sums over u32 arrays (device utility used between pipeline stages).
// Reduce-then-scan: 256 threads x 8 items per workgroup; the per-workgroup totals are scanned recursively.
#include <algorithm>

#include "internal.h"

namespace skh {

constexpr uint32_t SCAN_T = 256, SCAN_ITEMS = 8, SCAN_BLOCK = SCAN_T * SCAN_ITEMS;

// exclusive scan of one value per thread across the workgroup; *total = workgroup sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* lds /* >= 16 words */) {
    uint32_t incl = wave_incl_scan(v);
    uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 63) lds[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t i = 0; i < (blockDim.x >> 6); i++) { uint32_t t = lds[i]; if (i < w) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(256) void scan_reduce_kernel(const uint32_t* in, uint64_t n, uint32_t* block_sums) {
    __shared__ uint32_t lds[16];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (uint32_t i = 0; i < SCAN_ITEMS; i++) if (base + i < n) s += in[base + i];
    uint32_t tot; block_excl_scan(s, &tot, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void scan_down_kernel(const uint32_t* in, uint64_t n, const uint32_t* block_offs, uint32_t* out) {
    __shared__ uint32_t lds[16];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS]; uint32_t s = 0;
    for (uint32_t i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    uint32_t tot; uint32_t boff = block_offs[blockIdx.x];
    uint32_t off = block_excl_scan(s, &tot, lds) + boff;
    for (uint32_t i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = off; off += v[i]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = boff + tot;   // grand total
}

// up to SCAN_ONE_MAX values in ONE launch: a workgroup of 1024 threads walks the array 8192 values at a time and carries the running total (the scans between the
// pipeline's stages are over a few thousand to a few ten thousand values: three launches and a fill each were ~20 us of a 12.5 ms step, and there are four of them)
constexpr uint32_t SCAN_ONE_T = 1024, SCAN_ONE_MAX = 16 * SCAN_ONE_T * SCAN_ITEMS;
__global__ __launch_bounds__(1024) void scan_one_kernel(const uint32_t* in, uint32_t n, uint32_t* out) {
    __shared__ uint32_t lds[16];
    uint32_t carry = 0;
    for (uint32_t base0 = 0; base0 < n; base0 += SCAN_ONE_T * SCAN_ITEMS) {
        const uint32_t base = base0 + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS]; uint32_t s = 0;
        for (uint32_t i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
        uint32_t tot;
        uint32_t off = block_excl_scan(s, &tot, lds) + carry;
        for (uint32_t i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = off; off += v[i]; }
        carry += tot;
    }
    if (threadIdx.x == 0) out[n] = carry;
}

// several small regions set to a byte value in ONE launch (every hipMemsetAsync is a launch of its own: ~5 us each, and the chaining stage had eight of them in a row)
__global__ __launch_bounds__(256) void fill_regions_kernel(FillRegions fr) {
    uint32_t b = blockIdx.x, r = 0;
    while (r + 1 < fr.n && b >= fr.blocks[r]) { b -= fr.blocks[r]; r++; }
    uint32_t* p = fr.p[r]; const uint64_t words = fr.words[r]; const uint32_t v = fr.value[r];
    const uint64_t base = (uint64_t)b * (256 * 16);
#pragma unroll
    for (uint32_t u = 0; u < 16; u++) { const uint64_t x = base + (uint64_t)u * 256 + threadIdx.x; if (x < words) p[x] = v; }
}
void fill_regions(skh_ctx* ctx, FillRegions& fr) {
    uint32_t total = 0;
    for (uint32_t r = 0; r < fr.n; r++) { fr.blocks[r] = (uint32_t)((fr.words[r] + 256 * 16 - 1) / (256 * 16)); total += fr.blocks[r]; }
    if (!total) return;
    SKH_LAUNCH(fill_regions_kernel, total, 256, 0, ctx->stream, fr);
    check_launch("fill_regions");
}

// the reduce pass whose LAST workgroup (a ticket) scans the block totals itself: two launches for up to SCAN_TWO_MAX values.  The ticket counter belongs to the context
// (one per stream), is zero between scans and is put back to zero by the workgroup that used it up.
constexpr uint64_t SCAN_TWO_MAX = (uint64_t)65536 * SCAN_BLOCK;
__global__ __launch_bounds__(256) void scan_reduce_scan_kernel(const uint32_t* in, uint64_t n, uint32_t* block_sums, uint32_t* block_offs /* nb + 1 */, uint32_t nb, uint32_t* ticket) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t last;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (uint32_t i = 0; i < SCAN_ITEMS; i++) if (base + i < n) s += in[base + i];
    uint32_t tot; block_excl_scan(s, &tot, lds);
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = tot;
        __threadfence();
        last = atomicAdd(ticket, 1u) == nb - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
        const uint32_t b = b0 + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS]; uint32_t t = 0;
        for (uint32_t i = 0; i < SCAN_ITEMS; i++) { v[i] = b + i < nb ? __atomic_load_n(&block_sums[b + i], __ATOMIC_RELAXED) : 0u; t += v[i]; }
        uint32_t all;
        uint32_t off = block_excl_scan(t, &all, lds) + carry;
        for (uint32_t i = 0; i < SCAN_ITEMS; i++) { if (b + i < nb) block_offs[b + i] = off; off += v[i]; }
        carry += all;
    }
    if (threadIdx.x == 0) { block_offs[nb] = carry; __atomic_store_n(ticket, 0u, __ATOMIC_RELAXED); }
}

void exclusive_scan_u32(skh_ctx* ctx, const uint32_t* d_in, uint64_t n, uint32_t* d_out) {
    if (n == 0) { dzero(d_out, sizeof(uint32_t), ctx->stream); return; }
    if (n <= std::min<uint64_t>(SCAN_ONE_MAX, ctx->tune.scan_one_max)) {
        SKH_LAUNCH(scan_one_kernel, 1u, SCAN_ONE_T, 0, ctx->stream, d_in, (uint32_t)n, d_out);
        check_launch("scan_one");
        return;
    }
    uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (n <= std::min<uint64_t>(SCAN_TWO_MAX, ctx->tune.scan_two_max) && ctx->scan_ticket.p) {
        uint32_t* sums = ctx->arena.get<uint32_t>(nb);
        uint32_t* offs = ctx->arena.get<uint32_t>(nb + 1);
        uint32_t* ticket = ctx->scan_ticket.p + (ctx->stream == ctx->ring.s0 ? 0 : 1);
        SKH_LAUNCH(scan_reduce_scan_kernel, (unsigned)nb, SCAN_T, 0, ctx->stream, d_in, n, sums, offs, (uint32_t)nb, ticket);
        check_launch("scan_reduce_scan");
        SKH_LAUNCH(scan_down_kernel, (unsigned)nb, SCAN_T, 0, ctx->stream, d_in, n, (const uint32_t*)offs, d_out);
        check_launch("scan_down");
        return;
    }
    uint32_t* sums = ctx->arena.get<uint32_t>(nb);
    uint32_t* offs = ctx->arena.get<uint32_t>(nb + 1);
    SKH_LAUNCH(scan_reduce_kernel, (unsigned)nb, SCAN_T, 0, ctx->stream, d_in, n, sums);
    check_launch("scan_reduce");
    exclusive_scan_u32(ctx, sums, nb, offs);
    SKH_LAUNCH(scan_down_kernel, (unsigned)nb, SCAN_T, 0, ctx->stream, d_in, n, (const uint32_t*)offs, d_out);
    check_launch("scan_down");
}

}  // namespace skh
