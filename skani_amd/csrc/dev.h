// dev.h -- device runtime glue for the skani-hip kernels (gfx950).
//
// Product builds compile this with hipcc for gfx950 only.  The one conditional below lets the test suite compile the very same kernel sources against
// tests/emu/emu_dev.h, which gives every name of the HIP section a meaning on a lockstep CPU simulator (used to debug kernels in a container without
// a GPU); that switch is never defined in libskani_hip.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace skh {
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
struct PeerError : Error { using Error::Error; };          // a distributed call stops because ANOTHER rank failed (SKH_ERR_PEER): the caller's message is not the run's error
}

#ifdef SKANI_EMU
#include "emu_dev.h"
#else
// ================================================================================================ HIP (gfx950)
#include <hip/hip_runtime.h>
typedef hipStream_t devStream_t;
#define SKH_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (smem), (stream), __VA_ARGS__)
#define SKH_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]

namespace skh {

inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e));
}
// Device blocks go through a small caching allocator (alloc.hip): hipMalloc / hipFree of the few-hundred-MB sketch arrays
// cost 0.1-0.3 ms each and hipFree synchronises the device.  Every entry point of the library is synchronous, so a block is
// idle by the time its owner releases it.
void* dmalloc(size_t n);
void dfree(void* p);
void dcache_trim();                       // hand every cached block back to the driver
void dcache_stats(size_t* live_bytes, size_t* idle_bytes);   // device memory the library holds in use / idle in its cache (all devices of the process)
size_t device_memory_free();              // what the current device can still give: the driver's free bytes + this library's idle cached blocks
// Small uploads (offset tables, descriptors) go through a pinned ring that belongs to the CONTEXT the calling thread is working for (PinScope, set by
// every entry point of the C ABI): a copy from pageable memory is staged by the runtime and its first device read after the staging was measured
// at 130-150 us, in front of the kernels that wait for the table (rocpd timeline of a bench step); from pinned memory the copy is an ordinary
// asynchronous DMA and the caller's buffer is free at once.  A slot is reused only after the whole ring (8 MB of uploads) has gone by, and the
// wrap waits for the two streams of the ring's own context -- the only streams its slots are ever copied on.  The ring is freed with its context.
struct PinRing {
    char* p = nullptr; size_t cap = 0, off = 0; bool tried = false;
    hipStream_t s0 = nullptr, s1 = nullptr;                                          // the owning context's streams
    bool used0 = false, used1 = false;                                               // which of them has copied from / into a slot since the last wrap
    ~PinRing() { if (p) (void)hipHostFree(p); }
};
inline PinRing*& pin_ring_of_thread() { static thread_local PinRing* r = nullptr; return r; }
struct PinScope {                                                                    // entry points: this thread's device is `device`, its small uploads use `ring`
    PinRing* prev; int prev_dev = -1;                                                // (the caller's current device is put back on the way out: a host application may share the thread)
    PinScope(int device, PinRing* ring) : prev(pin_ring_of_thread()) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
        if (cur != device) { (void)hipSetDevice(device); prev_dev = cur; }
        pin_ring_of_thread() = ring;
    }
    ~PinScope() { pin_ring_of_thread() = prev; if (prev_dev >= 0) (void)hipSetDevice(prev_dev); }
};
// SKH_TUNE_PAGEABLE_DIRECT=1: large copies from / to pageable host memory go to the runtime as they are (the form before round 5; see d2h)
inline bool pageable_direct() { static const bool on = [] { const char* v = getenv("SKH_TUNE_PAGEABLE_DIRECT"); return v && *v && *v != '0'; }(); return on; }
inline bool pin_ring_ready(PinRing* r, devStream_t s) {
    constexpr size_t PIN_RING = (size_t)8 << 20;
    if (!r || !(s == r->s0 || s == r->s1)) return false;
    if (!r->tried) { r->tried = true; void* q = nullptr; if (hipHostMalloc(&q, PIN_RING, hipHostMallocPortable) == hipSuccess) { r->p = (char*)q; r->cap = PIN_RING; } else (void)hipGetLastError(); }
    return r->p != nullptr;
}
inline char* pin_ring_take(PinRing* r, size_t n, hipStream_t s) {                    // n <= cap / 2; a slot is reused only after the streams that used the ring since its last wrap have been waited for
    const size_t need = (n + 255) & ~(size_t)255;                                    // (only those: a read-back on the main stream must not sit out an exchange or an index sort the second stream is busy with)
    if (r->off + need > r->cap) {
        if (r->used0) hip_check(hipStreamSynchronize(r->s0), "pinned ring wrap");
        if (r->used1) hip_check(hipStreamSynchronize(r->s1), "pinned ring wrap");
        r->used0 = r->used1 = false; r->off = 0;
    }
    if (s == r->s0) r->used0 = true; else r->used1 = true;
    char* p = r->p + r->off; r->off += need; return p;
}
constexpr size_t PIN_CHUNK = (size_t)4 << 20;
inline void h2d(void* d, const void* h, size_t n, devStream_t s) {
    if (!n) return;
    constexpr size_t PIN_MAX = (size_t)1 << 20;
    PinRing* r = pin_ring_of_thread();
    if ((n <= PIN_MAX || !pageable_direct()) && pin_ring_ready(r, s)) {              // (larger than the ring's half: in chunks, the same way)
        for (size_t at = 0; at < n; at += PIN_CHUNK) {
            const size_t m = n - at < PIN_CHUNK ? n - at : PIN_CHUNK;
            char* q = pin_ring_take(r, m, s);
            memcpy(q, (const char*)h + at, m);
            hip_check(hipMemcpyAsync((char*)d + at, q, m, hipMemcpyHostToDevice, s), "h2d");
        }
        return;
    }
    hip_check(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s), "h2d");
}
inline void h2d_big(void* d, const void* h, size_t n, devStream_t s) { if (n) hip_check(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s), "h2d"); }   // large uploads from PINNED memory: asynchronous
// Device -> pageable host memory, synchronous.  A copy of more than a few pages into pageable memory makes the runtime pin the caller's buffer for the transfer and let go
// of the pin later, behind the caller's back -- and the FIRST KERNEL LAUNCH OF THE NEXT CALL then reached the device 20-40 ms late (round 5: a 5 MB read-back in one
// call, the next call's first launch 23-40 ms late, every time; this is the "start-of-process transient" of rounds 3-4, profiles/r05_first_steps_transient.md).  Such copies
// go through the context's pinned ring in 4 MB pieces instead: one more memcpy on the host, no pinning by the runtime.  (A destination that IS pinned: d2h_pinned.)
inline void d2h(void* h, const void* d, size_t n, devStream_t s) {
    if (!n) return;
    constexpr size_t DIRECT_MAX = (size_t)64 << 10;
    PinRing* r = pin_ring_of_thread();
    if (n > DIRECT_MAX && !pageable_direct() && pin_ring_ready(r, s)) {
        for (size_t at = 0; at < n; at += PIN_CHUNK) {
            const size_t m = n - at < PIN_CHUNK ? n - at : PIN_CHUNK;
            char* q = pin_ring_take(r, m, s);
            hip_check(hipMemcpyAsync(q, (const char*)d + at, m, hipMemcpyDeviceToHost, s), "d2h");
            hip_check(hipStreamSynchronize(s), "d2h sync");
            memcpy((char*)h + at, q, m);
        }
        return;
    }
    hip_check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "d2h"); hip_check(hipStreamSynchronize(s), "d2h sync");
}
inline void d2h_pinned(void* h, const void* d, size_t n, devStream_t s) { if (n) { hip_check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "d2h"); hip_check(hipStreamSynchronize(s), "d2h sync"); } }   // h: pinned memory
inline void d2h_async(void* h, const void* d, size_t n, devStream_t s) { if (n) hip_check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s), "d2h"); }   // h: pinned memory; the caller synchronises
inline void d2d(void* d, const void* s_, size_t n, devStream_t s) { if (n) hip_check(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s), "d2d"); }
inline void dzero(void* d, size_t n, devStream_t s) { if (n) hip_check(hipMemsetAsync(d, 0, n, s), "memset"); }
inline void dfill(void* d, int byte, size_t n, devStream_t s) { if (n) hip_check(hipMemsetAsync(d, byte, n, s), "memset"); }
inline void dsync(devStream_t s) { hip_check(hipStreamSynchronize(s), "stream sync"); }
inline void* pin_alloc(size_t n) { void* p = nullptr; hip_check(hipHostMalloc(&p, n ? n : 1, hipHostMallocPortable), "hipHostMalloc"); return p; }
inline void pin_free(void* p) { (void)hipHostFree(p); }
inline void check_launch(const char* what) { hip_check(hipGetLastError(), what); }
inline void device_sync_all() noexcept { (void)hipDeviceSynchronize(); }              // error paths: nothing queued may outlive the buffers that go away

// the device of a context and its two streams
inline void dev_open(int device, devStream_t* s0, devStream_t* s1) {
    int n = 0;
    hip_check(hipGetDeviceCount(&n), "hipGetDeviceCount");
    if (device < 0 || device >= n) throw Error("no such HIP device (this library has no CPU path)");
    hip_check(hipSetDevice(device), "hipSetDevice");
    hip_check(hipStreamCreateWithFlags(s0, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipStreamCreateWithFlags(s1, hipStreamNonBlocking), "hipStreamCreate");   // (stream priorities were measured: no effect on how the two share the GPU)
}
inline void dev_drain(int device, devStream_t s0, devStream_t s1) noexcept { (void)hipSetDevice(device); (void)hipStreamSynchronize(s0); (void)hipStreamSynchronize(s1); }
inline void dev_close(devStream_t s0, devStream_t s1) noexcept { (void)hipStreamDestroy(s0); (void)hipStreamDestroy(s1); }

// stream events (phase timings, the hand-over between the two streams)
struct DevEvent {
    hipEvent_t e = nullptr;
    DevEvent() { hip_check(hipEventCreate(&e), "hipEventCreate"); }
    ~DevEvent() { if (e) (void)hipEventDestroy(e); }
    DevEvent(const DevEvent&) = delete; DevEvent& operator=(const DevEvent&) = delete;
    DevEvent(DevEvent&& o) noexcept : e(o.e) { o.e = nullptr; }
    void record(devStream_t s) { hip_check(hipEventRecord(e, s), "hipEventRecord"); }
    void wait() { hip_check(hipEventSynchronize(e), "hipEventSynchronize"); }
    bool done() { const hipError_t r = hipEventQuery(e); if (r == hipSuccess) return true; (void)hipGetLastError(); return false; }
    void make_wait(devStream_t s) { hip_check(hipStreamWaitEvent(s, e, 0), "hipStreamWaitEvent"); }   // `s` continues after this event
    static float ms(const DevEvent& a, const DevEvent& b) { float t = 0; hip_check(hipEventElapsedTime(&t, a.e, b.e), "hipEventElapsedTime"); return t; }
};

// a kernel that asks for more than 64 KB of dynamic LDS has to be told so; the attribute belongs to the (device, function) pair, the call is cheap: made at every launch
template <class K> inline void kernel_allow_lds(K kernel, size_t bytes) {
    hip_check(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "LDS size attribute");
}

// A pointer that was put together from integers (broadcast through v_readlane, say) is a generic one to the compiler: its loads are flat_load, which count on two
// wait counters at once and make the compiler wait for everything in flight.  global_of() says what it is: global memory.
// (The host pass of the compiler only checks kernel bodies: it sees plain pointers.)
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> using GlobalPtr = const T __attribute__((address_space(1)))*;
#else
template <class T> using GlobalPtr = const T*;
#endif
template <class T> __device__ __forceinline__ GlobalPtr<T> global_of(const T* p) { return (GlobalPtr<T>)p; }

// ---- wave helpers (wave = 64 lanes on gfx950) ----
// uniform-lane broadcast: lane index is the same for the whole wave -> v_readlane on gfx950
__device__ __forceinline__ int wave_readlane(int v, int uniform_lane) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(uniform_lane)); }
// Point where lanes of ONE wave exchange data through LDS/global memory: orders the memory operations and
// keeps the compiler from moving accesses across it (the lanes themselves run in lockstep on hardware).
__device__ __forceinline__ void wave_sync_mem() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }
// Orders this thread's memory operations for the other threads of its WORKGROUP (use with __syncthreads()).  Not __threadfence(): an agent-scope
// fence on a multi-XCD part writes back / invalidates the XCD's L2 -- measured in round 2 at hundreds of microseconds per workgroup.
__device__ __forceinline__ void block_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// |a - b| of two unsigned numbers in one instruction (v_sad_u32; hipcc expands __usad to min/max/sub)
__device__ __forceinline__ uint32_t abs_diff_u32(uint32_t a, uint32_t b) { uint32_t d; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b)); return d; }
// value of the next / previous lane of the wave (lane 63 / lane 0 keep their own): one DPP move (wave_shl:1 / wave_shr:1) instead of a trip through the
// LDS crossbar -- for dependent chains of neighbour exchanges
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false); }
// inclusive prefix sum across the wave: four row shifts inside the rows of 16 lanes, then the two row broadcasts (lane 15 of a row to the next row, lane 31
// to the upper half) -- six DPP additions.  (Through __shfl_up it was six LDS-crossbar round trips plus a compare, a select and an address per step:
// ~40 vector instructions and six waits.)  (Every parity test runs through it: the seeding's hit lists and the join's anchor offsets are its sums.)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
    // v += v of the lane 1 / 2 / 4 / 8 below in the row (a lane without such a source keeps v: the instruction is off there), then rows 1, 3 += lane 15 of
    // the row before and rows 2, 3 += lane 31.  One instruction per step (through __builtin_amdgcn_update_dpp hipcc makes three); a DPP operand written by
    // the instruction before needs two idle states.
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}
// in-kernel timing (SKH_TRACE_JOIN): the shader clock, and a point at which a loaded value must have arrived
__device__ __forceinline__ unsigned long long wave_clock() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ void wait_for_value(uint32_t v) { asm volatile("" ::"v"(v)); }
// the XCD (0..7 on MI355X) this wave runs on, and an increment performed in that XCD's L2 (workgroup scope): screen.hip's per-XCD count planes
__device__ __forceinline__ uint32_t xcc_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xFu; }
// a word other CUs change with atomics while this kernel runs, read past this CU's L1 (screen.hip: the union-find's parents)
__device__ __forceinline__ uint32_t load_past_l1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void atomic_inc_xcd_local(uint32_t* p) { __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- the seeding loop's instructions (pack_seed.hip seed_tiles_kernel; the WHY is told there).  The simulator build has plain C++ under the same names (tests/emu/emu_dev.h).
#define SKH_SEED_DROP_NOTE() ((void)0)
// lanes with a >= b as a wave mask: ONE compare writing a scalar register pair (__ballot() goes through a select and a second compare)
__device__ __forceinline__ unsigned long long wave_mask_ge(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 35 /* ICMP_UGE */); }
// v |= bits in the lanes of the (wave-uniform, non-empty) mask: ONE vector instruction under a narrowed exec mask (as an `if` it is compare + select + or)
__device__ __forceinline__ void or_in_lanes(uint32_t& v, unsigned long long lanes, uint32_t bits) {
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %2\n\tv_or_b32 %0, %3, %0\n\ts_mov_b64 exec, %1" : "+v"(v), "=&s"(save) : "s"(lanes), "s"(bits));
}
__device__ __forceinline__ uint32_t funnel_shr(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
template <int SH> __device__ __forceinline__ uint64_t shl_add_u64(uint64_t a, uint64_t b) {          // (a << SH) + b, SH <= 4, one instruction
    uint64_t d; asm("v_lshl_add_u64 %0, %1, %3, %2" : "=v"(d) : "v"(a), "v"(b), "n"(SH)); return d;
}
// mm_hash64 (types.rs:86-96) of a 32-bit key in 16 instructions
__device__ __forceinline__ uint64_t seed_hash(uint32_t seed) {
    const uint64_t p = (uint64_t)seed * 0x200001ull;                   // key + (key << 21) < 2^54; the NOT of step 1 is folded into step 2:
    const uint32_t plo = (uint32_t)p, phi = (uint32_t)(p >> 32);       //   ~p ^ (~p >> 24) = p ^ (p >> 24) ^ 0xFFFFFF0000000000, and (phi >> 24) = 0
    const uint32_t lo2 = plo ^ __builtin_amdgcn_alignbit(phi, plo, 24);
    // step 3 (x 265): the high word of step 2, phi ^ 0xFFFFFF00 with phi < 2^22, is a small NEGATIVE number that fits a signed 24-bit operand,
    // so "hi * 265 + carry word of the low product" is a single 24-bit multiply-add
    const uint32_t hi2 = phi ^ 0xFFFFFF00u;
    const uint64_t q = (uint64_t)lo2 * 265u;
    uint32_t hi3; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(hi3) : "v"(hi2), "s"(265u), "v"((uint32_t)(q >> 32)));
    uint64_t key = ((uint64_t)hi3 << 32) | (uint32_t)q;
    key ^= key >> 14;
    key = shl_add_u64<4>(key, shl_add_u64<2>(key, key));               // x 21
    key ^= key >> 28;
    return shl_add_u64<0>(key << 31, key);
}
// the same mix up to its last step; returns n = ~hi + ((~hi:~lo) >> 1) of the key before that step, which is ~(hi + ((hi:lo) >> 1) + 1): the complement of
// "high word of key + (key << 31), carry of the low words taken as one".  ~x comes free: the step before is an XOR, taken as XNOR.
__device__ __forceinline__ uint32_t seed_probe(uint32_t seed) {
    const uint64_t p = (uint64_t)seed * 0x200001ull;
    const uint32_t plo = (uint32_t)p, phi = (uint32_t)(p >> 32);
    const uint32_t lo2 = plo ^ __builtin_amdgcn_alignbit(phi, plo, 24);
    const uint32_t hi2 = phi ^ 0xFFFFFF00u;
    const uint64_t q = (uint64_t)lo2 * 265u;
    uint32_t hi3; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(hi3) : "v"(hi2), "s"(265u), "v"((uint32_t)(q >> 32)));
    uint64_t key = ((uint64_t)hi3 << 32) | (uint32_t)q;
    key ^= key >> 14;
    key = shl_add_u64<4>(key, shl_add_u64<2>(key, key));               // x 21
    uint64_t sh; asm("v_lshrrev_b64 %0, 28, %1" : "=v"(sh) : "v"(key));   // (one instruction for both words; hipcc splits the shift into two)
    uint32_t nlo, nhi;
    asm("v_xnor_b32 %0, %1, %2" : "=v"(nlo) : "v"((uint32_t)key), "v"((uint32_t)sh));
    asm("v_xnor_b32 %0, %1, %2" : "=v"(nhi) : "v"((uint32_t)(key >> 32)), "v"((uint32_t)(sh >> 32)));
    return nhi + __builtin_amdgcn_alignbit(nhi, nlo, 1);
}

}  // namespace skh
#endif

namespace skh {

// grow-only pinned host buffer (descriptor arrays that are built on the host and copied to the device as they are: no staging copy)
struct PinBuf {
    void* p = nullptr; size_t cap = 0;
    ~PinBuf() { if (p) pin_free(p); }
    bool holds(const void* q) const { return p && (const char*)q >= (const char*)p && (const char*)q < (const char*)p + cap; }
    void* need(size_t bytes) { if (bytes > cap) { if (p) pin_free(p); p = nullptr; cap = 0; p = pin_alloc(bytes + bytes / 2); cap = bytes + bytes / 2; } return p; }
};

// RAII device buffer
template <class T> struct DBuf {
    T* p = nullptr; size_t n = 0;
    DBuf() {}
    explicit DBuf(size_t n_) { alloc(n_); }
    DBuf(const DBuf&) = delete; DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf& operator=(DBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DBuf() { release(); }
    void alloc(size_t n_) { release(); p = (T*)dmalloc(n_ * sizeof(T)); n = n_; }   // (an allocation that throws leaves the buffer empty)
    void release() { if (p) dfree(p); p = nullptr; n = 0; }
    size_t bytes() const { return n * sizeof(T); }
};

// ---- wave helpers built on the above ----
__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
template <class T> __device__ __forceinline__ T wave_bcast(T v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ unsigned wave_readlane(unsigned v, int uniform_lane) { return (unsigned)wave_readlane((int)v, uniform_lane); }
__device__ __forceinline__ uint64_t wave_readlane(uint64_t v, int uniform_lane) {
    return ((uint64_t)wave_readlane((unsigned)(v >> 32), uniform_lane) << 32) | wave_readlane((unsigned)v, uniform_lane);
}
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

}  // namespace skh
