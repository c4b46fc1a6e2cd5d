/*
 * emu.h -- TEST-ONLY lockstep simulator for the HIP kernels in skani_amd/csrc.
 *
 * There is no GPU in the build container and GPU minutes are rationed, so the kernel sources are also
 * compiled with g++ against this header (-DSKANI_EMU) into tests/emu/libskani_emu.so.  Every GPU thread
 * becomes a fiber; a workgroup runs on one OS thread; __syncthreads()/wave intrinsics are barriers between
 * fibers.  It exists to catch indexing/logic bugs and hangs before a kernel touches real hardware.
 *
 * It is NOT a product path and NOT a fallback: the skani_amd package never loads it (it loads
 * libskani_hip.so and fails loudly if that is missing); only `-m "not gpu"` tests named *_emu use it, and
 * no parity claim is made from it -- parity claims come from the `-m gpu` tests on an MI355X.
 */
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/mman.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint3_ { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct ulonglong2 { unsigned long long x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

namespace emu {

struct Fiber {
    void* sp = nullptr;          // saved stack pointer
    char* stack = nullptr;
    bool done = false;
    uint3_ tid{0, 0, 0};
    unsigned flat = 0;
};

struct WaveState { uint64_t buf[64]; unsigned arrived = 0; unsigned gen = 0; unsigned alive = 0; };

struct Block {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    unsigned n = 0, alive = 0, bar_arrived = 0, bar_gen = 0;
    uint3_ bid{0, 0, 0}; dim3 bdim, gdim;
    void* sched_sp = nullptr;
    unsigned cur = 0;
    const std::function<void()>* body = nullptr;
    char* dyn_smem = nullptr;
};

extern thread_local Block* g_blk;

extern "C" void emu_switch(void** save_sp, void* new_sp);

inline Fiber& cur() { return g_blk->fibers[g_blk->cur]; }
inline void yield() { Block* b = g_blk; emu_switch(&b->fibers[b->cur].sp, b->sched_sp); }

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
template <class K, class... A> inline void launch_k(dim3 grid, dim3 block, size_t smem, K kernel, A... args) {
    std::function<void()> body = [=]() { kernel(args...); };
    launch(grid, block, smem, body);
}

inline void syncthreads() {
    Block* b = g_blk; unsigned gen = b->bar_gen;
    if (++b->bar_arrived >= b->alive) { b->bar_arrived = 0; b->bar_gen++; }
    else while (b->bar_gen == gen) yield();
}
inline WaveState& wave() { return g_blk->waves[cur().flat >> 6]; }
inline void wave_sync() {
    WaveState& w = wave(); unsigned gen = w.gen;
    if (++w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
    else while (w.gen == gen) yield();
}
inline unsigned lane() { return cur().flat & 63; }

template <class T> inline T shfl(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl width");
    WaveState& w = wave(); uint64_t x = 0; memcpy(&x, &v, sizeof(T)); w.buf[lane()] = x;
    wave_sync(); uint64_t y = w.buf[src & 63]; wave_sync();
    T r; memcpy(&r, &y, sizeof(T)); return r;
}
inline uint64_t ballot(int pred) {
    WaveState& w = wave(); w.buf[lane()] = pred ? 1 : 0;
    wave_sync();
    uint64_t m = 0; unsigned base = (cur().flat >> 6) << 6;
    for (unsigned l = 0; l < 64 && base + l < g_blk->n; l++) if (!g_blk->fibers[base + l].done && w.buf[l]) m |= 1ull << l;
    wave_sync(); return m;
}
template <class T> inline T readfirstlane(T v) {
    unsigned base = (cur().flat >> 6) << 6; int first = 0;
    for (unsigned l = 0; l < 64 && base + l < g_blk->n; l++) if (!g_blk->fibers[base + l].done) { first = (int)l; break; }
    return shfl(v, first);
}

}  // namespace emu

#define threadIdx (emu::cur().tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)
#define warpSize 64

inline void __syncthreads() { emu::syncthreads(); }
template <class T> inline T __shfl(T v, int src, int = 64) { return emu::shfl(v, src); }
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) { int l = (int)emu::lane(); T r = emu::shfl(v, l - (int)d < 0 ? l : l - (int)d); return r; }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) { int l = (int)emu::lane(); T r = emu::shfl(v, l + (int)d > 63 ? l : l + (int)d); return r; }
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return emu::shfl(v, (int)emu::lane() ^ m); }
inline unsigned long long __ballot(int p) { return emu::ballot(p); }
inline int __any(int p) { return emu::ballot(p) != 0; }
inline int __all(int p) { return emu::ballot(!p) == 0; }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
inline unsigned long long __brevll(unsigned long long x) { unsigned long long r = 0; for (int i = 0; i < 64; i++) r |= ((x >> i) & 1ull) << (63 - i); return r; }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicMax(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
