// emu.cpp -- fiber scheduler of the TEST-ONLY kernel simulator (see emu.h).
#include "emu.h"

namespace emu {

thread_local Block* g_blk = nullptr;

__asm__(
    ".text\n.globl emu_switch\n.type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");

static constexpr size_t STACK = 256 * 1024;
static constexpr size_t MAX_THREADS = 1024;
static std::mutex g_pool_mu;
static std::vector<char*> g_pool;

static char* arena_get() {
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        if (!g_pool.empty()) { char* a = g_pool.back(); g_pool.pop_back(); return a; }
    }
    void* p = mmap(nullptr, STACK * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("emu mmap"); abort(); }
    return (char*)p;
}
static void arena_put(char* a) { std::lock_guard<std::mutex> l(g_pool_mu); g_pool.push_back(a); }

static void fiber_entry() {
    Block* b = g_blk;
    (*b->body)();
    Fiber& f = b->fibers[b->cur];
    f.done = true;
    b->alive--;
    WaveState& w = b->waves[f.flat >> 6];
    w.alive--;
    if (w.alive > 0 && w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
    if (b->alive > 0 && b->bar_arrived >= b->alive) { b->bar_arrived = 0; b->bar_gen++; }
    for (;;) emu_switch(&f.sp, b->sched_sp);
}

static void run_block(Block& b, char* arena) {
    g_blk = &b;
    unsigned n = b.n;
    for (unsigned i = 0; i < n; i++) {
        Fiber& f = b.fibers[i];
        f.done = false; f.flat = i;
        f.tid.x = i % b.bdim.x; f.tid.y = (i / b.bdim.x) % b.bdim.y; f.tid.z = i / (b.bdim.x * b.bdim.y);
        char* top = arena + (size_t)(i + 1) * STACK;
        uint64_t* sp = (uint64_t*)(top - 64);
        for (int k = 0; k < 6; k++) sp[k] = 0;
        sp[6] = (uint64_t)(void*)&fiber_entry;
        sp[7] = 0;
        f.sp = sp;
    }
    b.alive = n; b.bar_arrived = 0; b.bar_gen = 0;
    for (auto& w : b.waves) { w.arrived = 0; w.gen = 0; w.alive = 0; }
    for (unsigned i = 0; i < n; i++) b.waves[i >> 6].alive++;
    unsigned long long spins = 0;
    while (b.alive > 0) {
        for (unsigned i = 0; i < n; i++) {
            if (b.fibers[i].done) continue;
            b.cur = i;
            emu_switch(&b.sched_sp, b.fibers[i].sp);
        }
        if (++spins > 200000000ull) { fprintf(stderr, "emu: kernel appears hung (block %u)\n", b.bid.x); abort(); }
    }
    g_blk = nullptr;
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    unsigned nthr = block.x * block.y * block.z;
    if (nthr == 0 || nblocks == 0) return;
    if (nthr > MAX_THREADS) { fprintf(stderr, "emu: block too large\n"); abort(); }
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 4;
    const char* env = getenv("SKANI_EMU_THREADS"); if (env) hw = (unsigned)atoi(env);
    unsigned T = (unsigned)std::min<size_t>(hw, nblocks);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        char* arena = arena_get();
        Block b; b.n = nthr; b.bdim = block; b.gdim = grid; b.body = &body;
        b.fibers.resize(nthr); b.waves.resize((nthr + 63) / 64);
        std::vector<char> dyn(smem + 16);
        b.dyn_smem = dyn.data();
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bid.x = (unsigned)(i % grid.x); b.bid.y = (unsigned)((i / grid.x) % grid.y); b.bid.z = (unsigned)(i / ((size_t)grid.x * grid.y));
            run_block(b, arena);
        }
        arena_put(arena);
    };
    if (T <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

}  // namespace emu
