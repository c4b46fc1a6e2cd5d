// alloc.hip -- caching allocator under dmalloc/dfree (device memory for sketch sets, genome sets and the scratch arena).
//
// Freed blocks are kept, per device, and handed out again to requests of (nearly) the same size: a pipeline that sketches
// and compares batch after batch asks for the same array sizes every time.  The cache is bounded (SKH_TUNE_ALLOC_CACHE_BYTES,
// default 32 GiB per device; 0 disables it) and is flushed before an allocation is allowed to fail.
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

#include "dev.h"

namespace skh {

namespace {

struct DeviceCache { std::multimap<size_t, void*> idle; size_t idle_bytes = 0; };
struct Pool {
    std::mutex mu;
    std::unordered_map<void*, std::pair<size_t, int>> live;   // block -> (size, device)
    std::map<int, DeviceCache> dev;
    size_t cap;
    Pool() { const char* v = getenv("SKH_TUNE_ALLOC_CACHE_BYTES"); cap = v && *v ? (size_t)strtoull(v, nullptr, 10) : ((size_t)32 << 30); }
    void flush(DeviceCache& dc) { for (auto& kv : dc.idle) (void)hipFree(kv.second); dc.idle.clear(); dc.idle_bytes = 0; }
};
Pool& pool() { static Pool* p = new Pool(); return *p; }   // leaked on purpose: outlives every static DBuf

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
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) { (void)hipGetLastError(); P.flush(dc); e = hipMalloc(&p, n); }
    hip_check(e, "hipMalloc");
    P.live[p] = {n, device};
    return p;
}

void dfree(void* p) {
    if (!p) return;
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) { (void)hipFree(p); return; }
    const size_t sz = it->second.first; const int device = it->second.second;
    P.live.erase(it);
    DeviceCache& dc = P.dev[device];
    if (sz > P.cap || dc.idle_bytes + sz > P.cap) { (void)hipFree(p); return; }
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

void dcache_trim() {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto& kv : P.dev) P.flush(kv.second);
}

}  // namespace skh
