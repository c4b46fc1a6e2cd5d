// Issue-rate microbenchmark for the integer VALU instructions the seeding hash can be built from (gfx950).
// Each kernel runs ITER iterations of 8 independent chains of one instruction; 8 waves per SIMD; reports cycles per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/valu_rates tools/exp/valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITER 4096
#define CHAINS 8
#define DEF32(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {                       \
        uint32_t a[CHAINS];                                                                                \
        for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x * 17 + c;                               \
        uint32_t b = seed | 3;                                                                             \
        for (int i = 0; i < ITER; i++) {                                                                   \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(a[c]) : "v"(b) : "vcc");    \
        }                                                                                                  \
        uint32_t s = 0; for (int c = 0; c < CHAINS; c++) s ^= a[c];                                        \
        if (s == 0x12345) out[0] = s;                                                                      \
    }
#define DEF64(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {                       \
        uint64_t a[CHAINS];                                                                                \
        for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x * 17 + c;                               \
        uint32_t b = seed | 3;                                                                             \
        for (int i = 0; i < ITER; i++) {                                                                   \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(a[c]) : "v"(b) : "vcc");    \
        }                                                                                                  \
        uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s ^= a[c];                                        \
        if (s == 0x12345) out[0] = (uint32_t)s;                                                            \
    }
DEF32(xor, "v_xor_b32 %0, %0, %1")
DEF32(mov_chain, "v_mov_b32 %0, %0")
DEF32(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
DEF32(add3, "v_add3_u32 %0, %0, %1, %1")
DEF32(alignbit, "v_alignbit_b32 %0, %0, %1, 24")
DEF32(bfe, "v_bfe_u32 %0, %0, 2, 30")
DEF32(mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF32(mul_hi, "v_mul_hi_u32 %0, %0, %1")
DEF32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %1")
DEF32(bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96")
DEF32(min_u32, "v_min_u32 %0, %0, %1")
DEF32(add_co_pair, "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF32(cmp_cnd, "v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
DEF64(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
DEF64(mad_u64_u32_sdst, "v_mad_u64_u32 %0, s[10:11], %1, %1, %0")
DEF64(lshrrev_b64, "v_lshrrev_b64 %0, 7, %0")
DEF64(lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
DEF64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %0")
DEF64(cmp_u64, "v_cmp_gt_u64 vcc, %0, %0")
DEF64(mov_pair, "v_mov_b64 %0, %0")

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Case { const char* name; kern_t k; int instrs; };
int main() {
    uint32_t* d; hipMalloc(&d, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1e3;
    printf("device %s, %d CUs, clockRate %.0f MHz\n", p.gcnArchName, cus, mhz);
    std::vector<Case> cs = {
        {"v_xor_b32", k_xor, 1}, {"v_mov_b32", k_mov_chain, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_add3_u32", k_add3, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_bfe_u32", k_bfe, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1}, 
        {"v_bitop3_b32", k_bitop3, 1}, {"v_min_u32", k_min_u32, 1}, {"v_add_co+v_addc_co (pair)", k_add_co_pair, 2}, {"v_cmp_gt_u32+v_cndmask (pair)", k_cmp_cnd, 2},
        {"v_mad_u64_u32 (vcc)", k_mad_u64_u32, 1}, {"v_mad_u64_u32 (sgpr pair)", k_mad_u64_u32_sdst, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_cmp_gt_u64", k_cmp_u64, 1}, {"v_mov_b64", k_mov_pair, 1},
    };
    const int blocks = cus * 8;                    // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("| instruction | ms | cycles per wave-instruction per SIMD (at clockRate) |\n|---|---|---|\n");
    for (auto& c : cs) {
        c.k<<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 3; r++) { hipEventRecord(e0); c.k<<<blocks, 256>>>(d, 1); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        const double waves_per_simd = 8.0;
        const double instr_per_simd = waves_per_simd * ITER * CHAINS;      // groups (pairs count as one group)
        const double cyc = best * 1e-3 * mhz * 1e6 / instr_per_simd;
        printf("| %s | %.3f | %.2f |\n", c.name, best, cyc);
    }
    return 0;
}
