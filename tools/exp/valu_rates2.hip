// VALU issue-rate microbenchmark, second edition (gfx950): settles how many cycles a wave64 VALU instruction occupies a SIMD.
//
// Differences from valu_rates.hip (round 1), following the round-1 review:
//   * time is measured INSIDE the kernel, per wave, with both s_memtime (clock64) and the constant 100 MHz s_memrealtime (wall_clock64):
//     if the two differ, their ratio is the shader clock the kernel really ran at, and cycles are real cycles -- not "ms x nominal clockRate";
//   * fp32 controls (v_fma_f32 in VOP3 and v_fmac_f32 in VOP2 encoding, v_pk_fma_f32, v_add_f32, v_mul_f32) next to the integer instructions
//     of the seeding loop, and the same operation in both encodings (v_xor_b32 e32 / e64) to separate "operation" from "encoding";
//   * 1, 2, 4 and 8 waves per SIMD (grid = CUs x W workgroups of 4 waves; the placement is checked from HW_REG_HW_ID);
//   * every kernel is run three times and the last run is reported (clock ramp), in an order that interleaves cheap and expensive instructions.
// A second mode (argv[1] = "pmc") runs only W = 8, once per kernel, for a `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
// SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace` pass (tools/exp/valu_rates2.sh).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/valu_rates2 tools/exp/valu_rates2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
#define ITER 65536
#define CHAINS 8

struct WaveRec { uint64_t cyc, rt; uint32_t hwid, xcc; };

// The timed loop is long (524288 instructions per wave, 2-5 ms at 8 waves per SIMD) so that the ~40 us the dispatcher needs to place 8192 waves --
// during which the early waves have fewer than W - 1 neighbours -- stays below 2 % of a wave's life.  (A start barrier on an arrival counter was
// tried and is worse: 8192 pollers of one address leave the spin hundreds of microseconds apart.)
__device__ __forceinline__ void stamp(uint64_t& c, uint64_t& r) {
    c = __builtin_readcyclecounter();          // s_memtime
    r = wall_clock64();                        // s_memrealtime (100 MHz)
}
__device__ __forceinline__ void record(WaveRec* rec, uint64_t c0, uint64_t r0) {
    uint64_t c1, r1; stamp(c1, r1);
    if ((threadIdx.x & 63) == 0) {
        uint32_t hw, xc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
        rec[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = WaveRec{c1 - c0, r1 - r0, hw, xc};
    }
}

#define DEF32(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(WaveRec* rec, uint32_t* out, uint32_t seed) {         \
        uint32_t a[CHAINS];                                                                                \
        for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x * 17 + c;                               \
        uint32_t b = seed | 3;                                                                             \
        uint64_t c0, r0; stamp(c0, r0);                                                                    \
        for (int i = 0; i < ITER; i++) {                                                                   \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(a[c]) : "v"(b) : "vcc");    \
        }                                                                                                  \
        record(rec, c0, r0);                                                                               \
        uint32_t s = 0; for (int c = 0; c < CHAINS; c++) s ^= a[c];                                        \
        if (s == 0x12345) out[0] = s;                                                                      \
    }
#define DEF64(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(WaveRec* rec, uint32_t* out, uint32_t seed) {         \
        uint64_t a[CHAINS];                                                                                \
        for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x * 17 + c;                               \
        uint32_t b = seed | 3;                                                                             \
        uint64_t c0, r0; stamp(c0, r0);                                                                    \
        for (int i = 0; i < ITER; i++) {                                                                   \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(a[c]) : "v"(b) : "vcc");    \
        }                                                                                                  \
        record(rec, c0, r0);                                                                               \
        uint64_t s = 0; for (int c = 0; c < CHAINS; c++) s ^= a[c];                                        \
        if (s == 0x12345) out[0] = (uint32_t)s;                                                            \
    }
// fp32 controls
DEF32(fma_f32_vop3, "v_fma_f32 %0, %0, %1, %1")
DEF32(fmac_f32_vop2, "v_fmac_f32 %0, %1, %1")
DEF32(add_f32, "v_add_f32 %0, %0, %1")
DEF32(mul_f32, "v_mul_f32 %0, %0, %1")
DEF64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
DEF64(pk_add_f32, "v_pk_add_f32 %0, %0, %0")
// integer, 32-bit encodings (VOP1 / VOP2)
DEF32(xor_e32, "v_xor_b32 %0, %0, %1")
DEF32(xor_e64, "v_xor_b32_e64 %0, %0, %1")
DEF32(and_e32, "v_and_b32 %0, %0, %1")
DEF32(add_u32, "v_add_u32 %0, %0, %1")
DEF32(mov_e32, "v_mov_b32 %0, %0")
DEF32(lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
DEF32(min_u32, "v_min_u32 %0, %0, %1")
DEF32(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEF32(or_e32, "v_or_b32 %0, %0, %1")
DEF32(sub_u32, "v_sub_u32 %0, %0, %1")
DEF32(not_b32, "v_not_b32 %0, %0")
DEF32(lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
DEF32(max_u32, "v_max_u32 %0, %0, %1")
DEF32(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF32(addc_alone, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF32(xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEF32(add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD")
DEF32(mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF32(xor_dpp, "v_xor_b32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF32(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
// integer, VOP3 only
DEF32(and_or, "v_and_or_b32 %0, %0, %1, %1")
DEF32(xad, "v_xad_u32 %0, %0, %1, %1")
DEF32(or3, "v_or3_b32 %0, %0, %1, %1")
DEF32(perm, "v_perm_b32 %0, %0, %1, %1")
DEF32(bfi, "v_bfi_b32 %0, %0, %1, %1")
DEF32(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
DEF32(add3, "v_add3_u32 %0, %0, %1, %1")
DEF32(alignbit, "v_alignbit_b32 %0, %0, %1, 24")
DEF32(bfe, "v_bfe_u32 %0, %0, 2, 30")
DEF32(mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %1")
DEF32(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %1")
DEF32(bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96")
DEF32(sad_u32, "v_sad_u32 %0, %0, %1, 0")
DEF32(cmp_cnd, "v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
DEF32(addc_pair, "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc")
// 64-bit operands
DEF64(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
DEF64(lshrrev_b64, "v_lshrrev_b64 %0, 7, %0")
DEF64(lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
DEF64(lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %0")
DEF64(cmp_u64, "v_cmp_gt_u64 vcc, %0, %0")
DEF64(mov_b64, "v_mov_b64 %0, %0")
DEF64(pk_mov_b32, "v_pk_mov_b32 %0, %0, %0")

typedef void (*kern_t)(WaveRec*, uint32_t*, uint32_t);
struct Case { const char* name; const char* enc; kern_t k; int instrs; };

int main(int argc, char** argv) {
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1e3;
    std::vector<Case> cs = {
        {"v_fma_f32", "VOP3", k_fma_f32_vop3, 1}, {"v_xor_b32", "VOP2", k_xor_e32, 1}, {"v_mad_u64_u32", "VOP3", k_mad_u64_u32, 1}, {"v_fmac_f32", "VOP2", k_fmac_f32_vop2, 1},
        {"v_pk_fma_f32", "VOP3P", k_pk_fma_f32, 1}, {"v_xor_b32_e64", "VOP3", k_xor_e64, 1}, {"v_add_f32", "VOP2", k_add_f32, 1}, {"v_mul_f32", "VOP2", k_mul_f32, 1},
        {"v_pk_add_f32", "VOP3P", k_pk_add_f32, 1}, {"v_and_b32", "VOP2", k_and_e32, 1}, {"v_add_u32", "VOP2", k_add_u32, 1}, {"v_mov_b32", "VOP1", k_mov_e32, 1},
        {"v_lshlrev_b32", "VOP2", k_lshlrev_b32, 1}, {"v_min_u32", "VOP2", k_min_u32, 1}, {"v_mul_u32_u24", "VOP2", k_mul_u32_u24, 1},
        {"v_or_b32", "VOP2", k_or_e32, 1}, {"v_sub_u32", "VOP2", k_sub_u32, 1}, {"v_not_b32", "VOP1", k_not_b32, 1}, {"v_lshrrev_b32", "VOP2", k_lshrrev_b32, 1},
        {"v_max_u32", "VOP2", k_max_u32, 1}, {"v_cndmask_b32", "VOP2", k_cndmask, 1}, {"v_addc_co_u32", "VOP2", k_addc_alone, 1}, {"v_xor_b32_sdwa (src0 WORD_1)", "SDWA", k_xor_sdwa, 1},
        {"v_add_u32_sdwa (src0 BYTE_3)", "SDWA", k_add_sdwa, 1}, {"v_mov_b32_dpp row_shr:1", "DPP", k_mov_dpp, 1}, {"v_xor_b32_dpp row_shr:1", "DPP", k_xor_dpp, 1}, {"v_cvt_f32_u32", "VOP1", k_cvt_f32_u32, 1},
        {"v_and_or_b32", "VOP3", k_and_or, 1}, {"v_xad_u32", "VOP3", k_xad, 1}, {"v_or3_b32", "VOP3", k_or3, 1}, {"v_perm_b32", "VOP3", k_perm, 1}, {"v_bfi_b32", "VOP3", k_bfi, 1},
        {"v_lshl_add_u32", "VOP3", k_lshl_add_u32, 1}, {"v_add3_u32", "VOP3", k_add3, 1}, {"v_alignbit_b32", "VOP3", k_alignbit, 1}, {"v_bfe_u32", "VOP3", k_bfe, 1},
        {"v_mul_lo_u32", "VOP3", k_mul_lo, 1}, {"v_mad_u32_u24", "VOP3", k_mad_u32_u24, 1}, {"v_mad_i32_i24", "VOP3", k_mad_i32_i24, 1}, {"v_bitop3_b32", "VOP3", k_bitop3, 1},
        {"v_sad_u32", "VOP3", k_sad_u32, 1}, {"v_cmp_gt_u32 + v_cndmask_b32", "VOPC+VOP2", k_cmp_cnd, 2}, {"v_add_co_u32 + v_addc_co_u32", "VOP2 x2", k_addc_pair, 2},
        {"v_lshrrev_b64", "VOP3", k_lshrrev_b64, 1}, {"v_lshlrev_b64", "VOP3", k_lshlrev_b64, 1}, {"v_lshl_add_u64", "VOP3", k_lshl_add_u64, 1}, {"v_cmp_gt_u64", "VOPC", k_cmp_u64, 1},
        {"v_mov_b64", "VOP1", k_mov_b64, 1}, {"v_pk_mov_b32", "VOP3P", k_pk_mov_b32, 1},
    };
    WaveRec* d_rec; uint32_t* d_out;
    const int max_waves = cus * 8 * 4;
    hipMalloc(&d_rec, sizeof(WaveRec) * max_waves); hipMalloc(&d_out, 64); hipMemset(d_out, 0, 64);
    std::vector<WaveRec> rec(max_waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (pmc) {
        for (auto& c : cs) { hipMemset(d_out, 0, 64); c.k<<<cus * 8, 256>>>(d_rec, d_out, 1); hipDeviceSynchronize(); }
        printf("pmc mode: %zu kernels at 8 waves per SIMD, %d x %d = %d instructions per wave\n", cs.size(), ITER, CHAINS, ITER * CHAINS);
        return 0;
    }
    printf("device %s, %d CUs, clockRate %.0f MHz; %d x %d instructions per wave; W = waves per SIMD (grid = %d x W workgroups of 256 threads)\n", p.gcnArchName, cus, mhz,
           ITER, CHAINS, cus);
    printf("cyc/inst = elapsed s_memtime ticks of the slowest wave / (W x instructions per wave) (\"mean wave\": the same from the mean wave life -- lower, because a SIMD serves its oldest wave first); MHz = s_memtime ticks per 10 ns of s_memrealtime x 1000 (shader clock during the run);\n");
    printf("cyc@nominal = the round-1 figure: kernel time from hipEvents x nominal clockRate / (W x instructions per wave)\n\n");
    printf("| instruction | encoding | W=1 cyc/inst | W=2 | W=4 | W=8 | W=8, mean wave | MHz at W=8 | W=8 cyc@nominal | max waves on one SIMD at W=8 |\n|---|---|---|---|---|---|---|---|---|---|\n");
    for (auto& c : cs) {
        double cpi[4] = {0, 0, 0, 0}, mhz8 = 0, nominal8 = 0, mean8 = 0; int maxw8 = 0;
        const int Ws[4] = {1, 2, 4, 8};
        for (int wi = 0; wi < 4; wi++) {
            const int W = Ws[wi], blocks = cus * W, waves = blocks * 4;
            float ms = 0;
            for (int r = 0; r < 3; r++) {
                hipMemsetAsync(d_out, 0, 64); hipEventRecord(e0); c.k<<<blocks, 256>>>(d_rec, d_out, 1); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(rec.data(), d_rec, sizeof(WaveRec) * waves, hipMemcpyDeviceToHost);
            double sc = 0, sr = 0, mxc = 0; std::map<uint64_t, int> per_simd;
            for (int w = 0; w < waves; w++) {
                sc += (double)rec[w].cyc; sr += (double)rec[w].rt; mxc = (double)rec[w].cyc > mxc ? (double)rec[w].cyc : mxc;
                // HW_ID (gfx9): simd [5:4], cu [11:8], sh [12], se [15:13]; plus the XCC id
                const uint64_t key = ((uint64_t)(rec[w].xcc & 0xF) << 16) | (rec[w].hwid & 0xFF30u);
                per_simd[key]++;
            }
            int mx = 0; for (auto& kv : per_simd) mx = kv.second > mx ? kv.second : mx;
            const double n_inst = (double)ITER * CHAINS * c.instrs;
            cpi[wi] = mxc / (W * n_inst);                 // the LAST wave to finish: the SIMD arbitrates oldest-first, so the older waves of a SIMD run faster than their share
            if (W == 8) { mean8 = sc / waves / (W * n_inst); mhz8 = sc / sr * 100.0; nominal8 = ms * 1e-3 * mhz * 1e6 / (W * n_inst); maxw8 = mx; }
        }
        printf("| %s | %s | %.2f | %.2f | %.2f | %.2f | %.2f | %.0f | %.2f | %d |\n", c.name, c.enc, cpi[0], cpi[1], cpi[2], cpi[3], mean8, mhz8, nominal8, maxw8);
    }
    return 0;
}
