// does rocprim::radix_sort_keys honour begin_bit for small / mid-sized inputs?  sorts n random 64-bit keys on bits [32, 64) and on bits [0, 32) and counts inversions of the sorted field
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    for (size_t n : {100ul, 1200ul, 5000ul, 28000ul, 100000ul, 1000000ul, 5000000ul}) {
        std::vector<uint64_t> h(n); for (auto& x : h) x = rng();
        uint64_t *d_in, *d_out; hipMalloc(&d_in, n * 8); hipMalloc(&d_out, n * 8);
        hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice);
        size_t tb = 0; rocprim::radix_sort_keys(nullptr, tb, d_in, d_out, n, 32, 64, 0);
        void* tmp; hipMalloc(&tmp, tb ? tb : 16);
        hipError_t e = rocprim::radix_sort_keys(tmp, tb, d_in, d_out, n, 32, 64, 0);
        hipDeviceSynchronize();
        std::vector<uint64_t> o(n); hipMemcpy(o.data(), d_out, n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 1; i < n; i++) if ((o[i] >> 32) < (o[i - 1] >> 32)) bad++;
        // the same keys on bits [0, 32)
        e = rocprim::radix_sort_keys(tmp, tb, d_in, d_out, n, 0, 32, 0);
        hipDeviceSynchronize(); hipMemcpy(o.data(), d_out, n * 8, hipMemcpyDeviceToHost);
        size_t bad_lo = 0; for (size_t i = 1; i < n; i++) if ((uint32_t)o[i] < (uint32_t)o[i - 1]) bad_lo++;
        printf("n=%zu err=%d tmp=%zu inversions sorting bits [32,64): %zu   bits [0,32): %zu\n", n, (int)e, tb, bad, bad_lo);
        hipFree(d_in); hipFree(d_out); hipFree(tmp);
    }
}
