// scan.hip -- exclusive prefix sums over u32 arrays (device utility used between pipeline stages).
// Reduce-then-scan: 256 threads x 8 items per workgroup; the per-workgroup totals are scanned recursively.
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

void exclusive_scan_u32(skh_ctx* ctx, const uint32_t* d_in, uint64_t n, uint32_t* d_out) {
    if (n == 0) { dzero(d_out, sizeof(uint32_t), ctx->stream); return; }
    uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb == 1) {
        uint32_t* zero = ctx->arena.get<uint32_t>(1);
        dzero(zero, sizeof(uint32_t), ctx->stream);
        SKH_LAUNCH(scan_down_kernel, 1u, SCAN_T, 0, ctx->stream, d_in, n, (const uint32_t*)zero, d_out);
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
