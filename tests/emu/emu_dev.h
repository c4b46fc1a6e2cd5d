/*
 * emu_dev.h -- TEST-ONLY: what skani_amd/csrc/dev.h's HIP section means on the lockstep CPU simulator (emu.h).
 *
 * Included by dev.h in place of that section when the kernel sources are compiled with g++ -DSKANI_EMU into tests/emu/libskani_emu.so.  Every name the
 * product sources use from the HIP section has a plain-C++ meaning here: device memory is host memory, streams and events do nothing, the wave helpers
 * go through the simulator's fibers, the three inline-assembly helpers of the seeding loop (pack_seed.hip) keep their own conditional there.
 * Not a product path; the skani_amd package never loads the simulator build.
 */
#pragma once
#include "emu.h"

typedef int devStream_t;
#define SKH_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch_k(dim3(grid), dim3(block), (smem), kernel, __VA_ARGS__)
#define SKH_DYN_SMEM(name) char* name = emu::g_blk->dyn_smem

namespace skh {

inline void* dmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) throw Error("emu malloc failed"); return p; }
inline void dfree(void* p) { free(p); }
inline void dcache_trim() {}
inline void dcache_stats(size_t* live_bytes, size_t* idle_bytes) { if (live_bytes) *live_bytes = 0; if (idle_bytes) *idle_bytes = 0; }
inline size_t device_memory_free() { return (size_t)64 << 30; }
struct PinRing { devStream_t s0 = 0, s1 = 0; };
struct PinScope { PinScope(int, PinRing*) {} };
inline void h2d(void* d, const void* h, size_t n, devStream_t) { if (n) memcpy(d, h, n); }
inline void h2d_big(void* d, const void* h, size_t n, devStream_t) { if (n) memcpy(d, h, n); }
inline void d2h(void* h, const void* d, size_t n, devStream_t) { if (n) memcpy(h, d, n); }
inline void d2h_async(void* h, const void* d, size_t n, devStream_t) { if (n) memcpy(h, d, n); }
inline void d2h_pinned(void* h, const void* d, size_t n, devStream_t) { if (n) memcpy(h, d, n); }
inline void d2d(void* d, const void* s, size_t n, devStream_t) { if (n) memmove(d, s, n); }
inline void dzero(void* d, size_t n, devStream_t) { if (n) memset(d, 0, n); }
inline void dfill(void* d, int byte, size_t n, devStream_t) { if (n) memset(d, byte, n); }
inline void dsync(devStream_t) {}
inline void* pin_alloc(size_t n) { return malloc(n ? n : 1); }
inline void pin_free(void* p) { free(p); }
inline void check_launch(const char*) {}
inline void device_sync_all() noexcept {}
inline void dev_open(int, devStream_t* s0, devStream_t* s1) { *s0 = 0; *s1 = 0; }
inline void dev_drain(int, devStream_t, devStream_t) noexcept {}
inline void dev_close(devStream_t, devStream_t) noexcept {}
struct DevEvent {
    void record(devStream_t) {}
    void wait() {}
    bool done() { return true; }
    void make_wait(devStream_t) {}
    static float ms(const DevEvent&, const DevEvent&) { return 0.f; }
};
template <class K> inline void kernel_allow_lds(K, size_t) {}

template <class T> using GlobalPtr = const T*;
template <class T> inline GlobalPtr<T> global_of(const T* p) { return p; }
inline int wave_readlane(int v, int uniform_lane) { return emu::shfl(v, uniform_lane); }
inline void wave_sync_mem() { emu::wave_sync(); }
inline void block_fence() { __threadfence_block(); }
inline uint32_t abs_diff_u32(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }
inline uint32_t lane_next(uint32_t v) { const int l = (int)emu::lane(); return emu::shfl(v, l < 63 ? l + 1 : l); }
inline uint32_t lane_prev(uint32_t v) { const int l = (int)emu::lane(); return emu::shfl(v, l > 0 ? l - 1 : l); }
inline unsigned wave_incl_scan(unsigned v) {
    const unsigned l = emu::lane();
    for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(v, (unsigned)d, 64); if (l >= (unsigned)d) v += t; }
    return v;
}
inline unsigned long long wave_clock() { return 0; }
inline void wait_for_value(uint32_t) {}
inline uint32_t xcc_id() { return 0; }
inline void atomic_inc_xcd_local(uint32_t* p) { atomicAdd(p, 1u); }
inline uint32_t load_past_l1(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }

// ---- the seeding loop's instructions (dev.h has the gfx950 forms).  The mix is written out here on its own (types.rs:86-96): this header comes before common.h's mm_hash64.
inline uint64_t emu_mm_hash64(uint64_t key) {
    key = ~(key + (key << 21)); key ^= key >> 24; key = key + (key << 3) + (key << 8); key ^= key >> 14; key = key + (key << 2) + (key << 4); key ^= key >> 28; key += key << 31;
    return key;
}
inline uint32_t funnel_shr(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }
inline uint64_t seed_hash(uint32_t seed) { return emu_mm_hash64((uint64_t)seed); }
// (the simulator's superset is deliberately loose -- the hash's leading 16 bits -- so that the drop path runs in every test genome: ~0.1 % of its candidates are not hits)
inline uint32_t seed_probe(uint32_t seed) { return ~((uint32_t)(emu_mm_hash64((uint64_t)seed) >> 32) & 0xFFFF0000u); }
inline unsigned long long wave_mask_ge(uint32_t a, uint32_t b) { return __ballot(a >= b); }
inline void or_in_lanes(uint32_t& v, unsigned long long lanes, uint32_t bits) { if ((lanes >> (threadIdx.x & 63u)) & 1ull) v |= bits; }

}  // namespace skh
extern "C" { inline unsigned long long skh_emu_seed_drops = 0; }       // candidates the seeding kernel's dense pass dropped (the tests assert that the path runs)
#define SKH_SEED_DROP_NOTE() __atomic_fetch_add(&skh_emu_seed_drops, 1ull, __ATOMIC_RELAXED)
