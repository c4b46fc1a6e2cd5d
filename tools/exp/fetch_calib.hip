// Calibration of the TCC FETCH_SIZE / WRITE_SIZE counters on gfx950 for the two access patterns of this code base: wide coalesced streams and random
// 8-byte reads (one per 64-byte line, far apart: the join's table probes).  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (tools/exp/fetch_calib.sh).
// Every kernel reads a known number of bytes from a 4 GiB buffer (>> L2 + MALL), so the counter's unit per access pattern can be read off.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/fetch_calib tools/exp/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ __launch_bounds__(256) void k_stream16(const uint4* p, uint64_t n16, uint32_t* out) {          // n16 x 16 bytes, coalesced dwordx4
    uint32_t a = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) { const uint4 v = p[i]; a += v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345) out[0] = a;
}
__global__ __launch_bounds__(256) void k_stream4(const uint32_t* p, uint64_t n4, uint32_t* out) {          // n4 x 4 bytes, coalesced dword
    uint32_t a = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) a += p[i];
    if (a == 0x12345) out[0] = a;
}
__global__ __launch_bounds__(256) void k_random8(const uint64_t* p, uint64_t n_lines, uint64_t n_reads, uint32_t* out) {   // n_reads x 8 bytes, each in another random 64-byte line
    uint64_t a = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * 256) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        a += p[(x % n_lines) * 8 + (x >> 61)];
    }
    if (a == 0x12345) out[0] = (uint32_t)a;
}
__global__ __launch_bounds__(256) void k_write16(uint4* p, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void k_write4_scattered(uint32_t* p, uint64_t n_lines, uint64_t n_writes) {   // n_writes x 4 bytes, each in another random 64-byte line
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_writes; i += (uint64_t)gridDim.x * 256) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[(x % n_lines) * 16 + (x >> 60)] = (uint32_t)i;
    }
}
int main() {
    const uint64_t bytes = 4ull << 30;
    void* buf; uint32_t* out; hipMalloc(&buf, bytes); hipMalloc(&out, 64); hipMemset(buf, 1, bytes);
    const uint64_t n_lines = bytes / 64, n_reads = 1ull << 26;                        // 64 M random reads
    hipLaunchKernelGGL(k_stream16, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_stream4, dim3(8192), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out);
    hipLaunchKernelGGL(k_random8, dim3(8192), dim3(256), 0, 0, (const uint64_t*)buf, n_lines, n_reads, out);
    hipLaunchKernelGGL(k_write16, dim3(8192), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    hipLaunchKernelGGL(k_write4_scattered, dim3(8192), dim3(256), 0, 0, (uint32_t*)buf, n_lines, n_reads);
    hipDeviceSynchronize();
    printf("expected: k_stream16 %llu B, k_stream4 %llu B, k_random8 %llu reads (x 8 B requested, x 64 B lines = %llu B), k_write16 %llu B, k_write4_scattered %llu writes (x 4 B, x 64 B lines = %llu B)\n",
           (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)n_reads, (unsigned long long)(n_reads * 64), (unsigned long long)bytes, (unsigned long long)n_reads, (unsigned long long)(n_reads * 64));
    return 0;
}
