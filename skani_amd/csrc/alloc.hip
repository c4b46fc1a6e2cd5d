// alloc.hip -- caching allocator under dmalloc/dfree (device memory for sketch sets, genome sets and the scratch arena).
//
// Freed blocks are kept, per device, and handed out again to requests of (nearly) the same size: a pipeline that sketches
// and compares batch after batch asks for the same array sizes every time.  The cache is bounded (SKH_TUNE_ALLOC_CACHE_BYTES,
// default a third of the block's device, beyond 8 GiB only while an eighth of that device stays free; 0 disables it) and is flushed before an allocation is allowed to fail.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

#include "dev.h"

namespace skh {

namespace {

struct DeviceCache { std::multimap<size_t, void*> idle; size_t idle_bytes = 0; size_t cap = 0; bool cap_known = false; };
// the calling thread on `device` for the scope (memory queries answer for the current device), its previous device back afterwards
struct OnDevice {
    int prev = -1; bool moved = false;
    explicit OnDevice(int device) { if (hipGetDevice(&prev) == hipSuccess && prev != device) moved = hipSetDevice(device) == hipSuccess; else (void)hipGetLastError(); }
    ~OnDevice() { if (moved) (void)hipSetDevice(prev); }
};
struct Pool {
    std::mutex mu;
    std::unordered_map<void*, std::pair<size_t, int>> live;   // block -> (size, device)
    std::map<int, DeviceCache> dev;
    size_t env_cap = 0; bool cap_from_env = false, touch = false;
    Pool() {
        const char* v = getenv("SKH_TUNE_ALLOC_CACHE_BYTES"); if (v && *v) { env_cap = (size_t)strtoull(v, nullptr, 10); cap_from_env = true; }
        const char* t = getenv("SKH_TUNE_ALLOC_TOUCH"); touch = t && *t && *t != '0';
    }
    // default bound, per device: a third of ITS memory (96 GB on an MI355X).  A step over 10,000 genomes frees and asks again for ~45 GB of arrays; with the earlier 32 GiB
    // bound 9.4 GB of them went back to the driver and came from it again in every step, and the driver's clearing of fresh memory held the step's first kernel back by 10-20 ms.
    size_t bound(int device, DeviceCache& dc) {
        if (cap_from_env) return env_cap;
        if (!dc.cap_known) {
            OnDevice on(device); size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) dc.cap = tot / 3; else { (void)hipGetLastError(); dc.cap = (size_t)32 << 30; }
            dc.cap_known = true;
        }
        return dc.cap;
    }
    void flush(DeviceCache& dc) { for (auto& kv : dc.idle) (void)hipFree(kv.second); dc.idle.clear(); dc.idle_bytes = 0; }
};
Pool& pool() { static Pool* p = new Pool(); return *p; }   // leaked on purpose: outlives every static DBuf

// SKH_TRACE_ALLOC=1: every request that reaches the driver (a cache miss, a block the cache has no room for) is printed with its size and how long the call took
bool trace_alloc() { static const bool on = [] { const char* v = getenv("SKH_TRACE_ALLOC"); return v && *v && *v != '0'; }(); return on; }
struct DriverCall {
    const char* what; size_t bytes; std::chrono::steady_clock::time_point t0;
    DriverCall(const char* w, size_t b) : what(w), bytes(b) { if (trace_alloc()) t0 = std::chrono::steady_clock::now(); }
    ~DriverCall() { if (trace_alloc()) fprintf(stderr, "[skh alloc] %s %.1f MB  %.3f ms\n", what, bytes / 1048576.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};

}  // namespace

void* dmalloc(size_t n) {
    n = n ? (n + 255) & ~(size_t)255 : 256;
    int device = 0; hip_check(hipGetDevice(&device), "hipGetDevice");
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    DeviceCache& dc = P.dev[device];
    auto it = dc.idle.lower_bound(n);
    if (it != dc.idle.end() && it->first - n <= n / 8) {                            // at most 12.5 % slack
        void* p = it->second; const size_t sz = it->first;
        dc.idle.erase(it); dc.idle_bytes -= sz; P.live[p] = {sz, device};
        return p;
    }
    void* p = nullptr;
    DriverCall dc_trace("hipMalloc", n);
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) { (void)hipGetLastError(); P.flush(dc); e = hipMalloc(&p, n); }
    hip_check(e, "hipMalloc");
    // SKH_TUNE_ALLOC_TOUCH=1 (diagnosis of profiles/r04_first_steps_transient.md): a block fresh from the driver is written once here, so that whatever the driver defers
    // to a block's first use is paid at the allocation and not by the first kernel that touches it
    if (P.touch && n >= ((size_t)1 << 20)) { (void)hipMemset(p, 0, n); (void)hipDeviceSynchronize(); }
    P.live[p] = {n, device};
    return p;
}

void dfree(void* p) {
    if (!p) return;
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) { DriverCall t("hipFree (unknown block)", 0); (void)hipFree(p); return; }
    const size_t sz = it->second.first; const int device = it->second.second;
    P.live.erase(it);
    DeviceCache& dc = P.dev[device];
    const size_t cap = P.bound(device, dc);
    bool keep = sz <= cap && dc.idle_bytes + sz <= cap;
    // a cache beyond 8 GiB only while the block's device has room to spare for others (a resident database, the caller's own allocations, other processes: eight
    // ranks on one GPU each keep a cache of their own)
    if (keep && !P.cap_from_env && dc.idle_bytes + sz > ((size_t)8 << 30)) {
        OnDevice on(device); size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); keep = false; }
        else keep = fr >= tot / 8;
    }
    if (!keep) { DriverCall t("hipFree (cache full)", sz); (void)hipFree(p); return; }
    dc.idle.emplace(sz, p); dc.idle_bytes += sz;
}

void dcache_stats(size_t* live_bytes, size_t* idle_bytes) {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    size_t live = 0, idle = 0;
    for (auto& kv : P.live) live += kv.second.first;
    for (auto& kv : P.dev) idle += kv.second.idle_bytes;
    if (live_bytes) *live_bytes = live;
    if (idle_bytes) *idle_bytes = idle;
}

size_t device_memory_free() {
    int device = 0; size_t fr = 0, tot = 0;
    if (hipGetDevice(&device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return (size_t)16 << 30; }
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.dev.find(device);
    return fr + (it == P.dev.end() ? 0 : it->second.idle_bytes);
}

void dcache_trim() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto& kv : P.dev) P.flush(kv.second);
}

}  // namespace skh
