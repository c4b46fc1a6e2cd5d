// Does gfx950 accept more than 64 KB of LDS per workgroup?  (160 KB per CU.)  build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/lds_limit tools/exp/lds_limit.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned words) {
    extern __shared__ unsigned s[];
    for (unsigned i = threadIdx.x; i < words; i += 1024) s[i] = i;
    __syncthreads();
    unsigned a = 0; for (unsigned i = threadIdx.x; i < words; i += 1024) a += s[words - 1 - i];
    atomicAdd(out, a);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlockOptin %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    unsigned* d; hipMalloc(&d, 4);
    for (unsigned kb : {32u, 64u, 65u, 96u, 128u, 160u}) {
        hipMemset(d, 0, 4);
        hipError_t e0 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipLaunchKernelGGL(k, dim3(512), dim3(1024), kb * 1024, 0, d, kb * 256);
        hipError_t e1 = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
        unsigned h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%3u KB: setattr %s, launch %s, sync %s, sum %u\n", kb, hipGetErrorString(e0), hipGetErrorString(e1), hipGetErrorString(e2), h);
    }
    return 0;
}
