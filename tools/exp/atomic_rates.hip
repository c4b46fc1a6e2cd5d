// What does an increment in the L2 cost on gfx950, and does it matter how the lanes of one instruction fall onto cache lines?
// The triangle screen's count kernel (screen.hip) is 20 M returnless increments into ~9,500 hot cells x 8 per-XCD planes per step at 1,000 genomes.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/atomic_rates tools/exp/atomic_rates.hip        run: tools/exp/atomic_rates > gpurun_out/atomic_rates.txt
// Every kernel: 2^20 threads x R increments each; the target of increment r of thread t is a hash of (t, r) shaped by `mode`:
//   0 scattered : every lane its own random word among `words` hot words (what the count kernel does: lane = key, step = partner)
//   1 rows of 16: the 16 lanes of a quarter wave hit the 16 words of ONE random 64-byte line (lane = partner, one row of the matrix per instruction)
//   2 rows of 16, ragged: as 1, but only ~half of the 16 lanes take part (a clade member shares a marker with about half of its clade)
//   3 packed pairs: one 64-bit add of (1 | 1 << 32) on a random even word pair (two neighbouring cells in one operation)
//   4 same word: all 64 lanes the same random word (the worst case a marker shared by everybody would be, were it not one group)
// scope: 0 = workgroup-scope relaxed (stays in the XCD's L2: the per-XCD planes), 1 = agent scope (device-wide: one plane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xFu; }
template <int MODE, int SCOPE>
__global__ __launch_bounds__(256) void k_inc(uint32_t* cnt, uint32_t words /* power of two, per plane */, uint32_t planes, uint32_t R) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t* mine = cnt + (size_t)(planes > 1 ? xcc_id() % planes : 0) * words;
    for (uint32_t r = 0; r < R; r++) {
        uint32_t w; bool on = true;
        if (MODE == 0) w = mix(t * 64u + r) & (words - 1);
        else if (MODE == 1 || MODE == 2) { w = ((mix((t >> 4) * 64u + r) & (words - 1)) & ~15u) | (lane & 15u); if (MODE == 2) on = (mix(t * 64u + r + 77u) & 1u) != 0; }
        else if (MODE == 3) w = mix(t * 64u + r) & (words - 1) & ~1u;
        else w = mix((t >> 6) * 64u + r) & (words - 1);
        if (!on) continue;
        if (MODE == 3) {
            if (SCOPE) __hip_atomic_fetch_add((unsigned long long*)(mine + w), 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add((unsigned long long*)(mine + w), 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (SCOPE) __hip_atomic_fetch_add(mine + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(mine + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
template <int MODE, int SCOPE> static void run(const char* name, uint32_t* d, uint32_t words, uint32_t planes, size_t bytes) {
    const uint32_t T = 1u << 20, R = 20;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f; uint64_t sum = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(d, 0, bytes);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_inc<MODE, SCOPE>), dim3(T / 256), dim3(256), 0, 0, d, words, planes, R);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    std::vector<uint32_t> h((size_t)words * planes); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (uint32_t v : h) sum += v;
    const double ops = MODE == 3 ? (double)T * R : (double)sum;              // (operations issued per lane that took part; packed pairs: one operation, two cells)
    printf("| %-22s | %-9s | %7u x %u | %8.3f | %7.1f | %7.1f | %llu |\n", name, SCOPE ? "agent" : "workgroup", words, planes, best, ops / best / 1e6, (double)sum / best / 1e6, (unsigned long long)sum);
}
int main() {
    const size_t bytes = (size_t)8 << 22 << 2;   // up to 4 M words x 8 planes
    uint32_t* d; hipMalloc(&d, bytes);
    printf("| pattern | scope | hot words x planes | ms (best of 4) | G lane-operations/s | G cell increments/s | increments counted |\n|---|---|---|---|---|---|---|\n");
    for (uint32_t words : {16384u, 131072u, 4194304u}) {
        run<0, 0>("scattered", d, words, 8, bytes);            run<0, 1>("scattered", d, words, 1, bytes);
        run<1, 0>("16 lanes = one line", d, words, 8, bytes);  run<1, 1>("16 lanes = one line", d, words, 1, bytes);
        run<2, 0>("~8 of 16 lanes, one line", d, words, 8, bytes); run<2, 1>("~8 of 16 lanes, one line", d, words, 1, bytes);
        run<3, 0>("packed pair (64-bit)", d, words, 8, bytes);  run<3, 1>("packed pair (64-bit)", d, words, 1, bytes);
        run<4, 0>("64 lanes one word", d, words, 8, bytes);    run<4, 1>("64 lanes one word", d, words, 1, bytes);
    }
    return 0;
}
