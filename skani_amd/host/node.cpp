// node.cpp -- the GPUs of one node behind `skani-hip triangle --gpus N`: a launcher that forks one process per GPU and the ranks' host-side collectives.
//
// The reference's triangle uses every core it is given from the one command (triangle.rs:55-105, `-t`); the one command here uses every GPU it is given.  The
// launcher forks the ranks BEFORE anything touches the HIP runtime (a forked HIP context is unusable), waits for them, and when one fails stops the others: a rank
// that waits in a collective for a peer that has died would wait for ever.  The ranks share
//   * a control block (anonymous shared memory made before the fork): a barrier, the failure flag, the counts / offsets of the all-to-all, the RCCL unique id;
//   * one data segment per rank (memfd made before the fork, so every rank holds every descriptor): a rank writes what it sends into its own segment, grows it
//     when needed and publishes the size; its peers map it (again, when it grew) and copy their part out.
// These collectives carry the control plane in every mode (file shares, genome names for rank 0's writers, the agreement on the transport) and are the DATA plane's
// fall-back -- skh_comm_create_host over them -- when RCCL is not usable (all ranks on one device: `--one-device`, which is how the path runs under test on a
// one-GPU box; or a communicator that fails its self-test).
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <climits>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "host.hpp"

namespace skhost {

namespace {
constexpr int MAXW = 64;
struct Ctrl {
    std::atomic<uint32_t> arrived, generation, failed;                                // failed: 0 = no, otherwise 1 + the rank that claimed the failure (or 1 + MAXW: the launcher)
    std::atomic<uint64_t> seg_bytes[MAXW];                                          // size of every rank's data segment
    uint64_t cnt[MAXW][MAXW], off[MAXW][MAXW];                                      // all-to-all: [source][destination]
};
long futex(std::atomic<uint32_t>* a, int op, uint32_t val, const timespec* ts) { return syscall(SYS_futex, (uint32_t*)a, op, val, ts, nullptr, 0); }
}  // namespace

struct Node {
    int world = 1, rank = -1;
    Ctrl* c = nullptr;
    int fd[MAXW]; uint8_t* map[MAXW]; uint64_t mapped[MAXW];
};

Node* node_create(int world) {
    if (world < 1 || world > MAXW) throw std::runtime_error("--gpus: between 1 and 64 ranks");
    Node* n = new Node(); n->world = world;
    void* p = mmap(nullptr, sizeof(Ctrl), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) throw std::runtime_error("cannot map the ranks' control block");
    n->c = new (p) Ctrl();
    for (int r = 0; r < MAXW; r++) { n->fd[r] = -1; n->map[r] = nullptr; n->mapped[r] = 0; n->c->seg_bytes[r] = 0; }
    for (int r = 0; r < world; r++) {
        char name[32]; snprintf(name, sizeof name, "skani-hip-rank%d", r);
        n->fd[r] = (int)syscall(SYS_memfd_create, name, 0u);
        if (n->fd[r] < 0) throw std::runtime_error("memfd_create failed");
    }
    return n;
}
int node_rank(const Node* n) { return n->rank; }
int node_world(const Node* n) { return n->world; }

// true for exactly one caller: the one whose message is the error of the run
bool node_claim_failure(Node* n) {
    uint32_t none = 0;
    const bool first = n->c->failed.compare_exchange_strong(none, 1u + (uint32_t)(n->rank < 0 ? MAXW : n->rank));
    n->c->generation.fetch_add(1); futex(&n->c->generation, FUTEX_WAKE, INT_MAX, nullptr);     // whoever waits in a barrier looks at the flag again
    return first;
}
bool node_failed(const Node* n) { return n->c->failed.load() != 0; }

bool node_barrier(Node* n) {
    Ctrl* c = n->c;
    if (c->failed.load()) return false;
    const uint32_t gen = c->generation.load(std::memory_order_acquire);
    if (c->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)n->world) {
        c->arrived.store(0, std::memory_order_relaxed);
        c->generation.fetch_add(1, std::memory_order_release);
        futex(&c->generation, FUTEX_WAKE, INT_MAX, nullptr);
        return !c->failed.load();
    }
    for (uint32_t spin = 0;; spin++) {
        if (c->failed.load()) return false;
        if (c->generation.load(std::memory_order_acquire) != gen) return true;
        if (spin < 4000) { __builtin_ia32_pause(); continue; }
        const timespec ts{0, 20 * 1000 * 1000};                                     // (the time-out only bounds how late a failure is noticed)
        futex(&c->generation, FUTEX_WAIT, gen, &ts);
    }
}

namespace {
bool remap(Node* n, int r, uint64_t bytes) {
    if (n->map[r]) munmap(n->map[r], n->mapped[r]);
    n->map[r] = nullptr; n->mapped[r] = 0;
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, n->fd[r], 0);
    if (p == MAP_FAILED) return false;
    n->map[r] = (uint8_t*)p; n->mapped[r] = bytes;
    return true;
}
// this rank's segment holds `bytes` bytes of `src`
bool publish(Node* n, const void* src, uint64_t bytes) {
    const int me = n->rank;
    if (bytes > n->mapped[me]) {
        const uint64_t cap = (std::max<uint64_t>(bytes, 2 * n->mapped[me]) + 0xFFFFFull) & ~0xFFFFFull;
        if (ftruncate(n->fd[me], (off_t)cap) != 0 || !remap(n, me, cap)) return false;
        n->c->seg_bytes[me].store(cap, std::memory_order_release);
    }
    if (bytes) memcpy(n->map[me], src, bytes);
    return true;
}
const uint8_t* peer(Node* n, int r, uint64_t need) {
    if (need <= n->mapped[r]) return n->map[r];
    const uint64_t cap = n->c->seg_bytes[r].load(std::memory_order_acquire);
    if (cap < need || !remap(n, r, cap)) return nullptr;
    return n->map[r];
}
}  // namespace

bool node_all_gather(Node* n, const void* send, void* recv, uint64_t bytes) {
    if (!publish(n, send, bytes)) { node_claim_failure(n); return false; }
    if (!node_barrier(n)) return false;
    for (int r = 0; r < n->world && bytes; r++) {
        const uint8_t* p = peer(n, r, bytes);
        if (!p) { node_claim_failure(n); return false; }
        memcpy((uint8_t*)recv + (uint64_t)r * bytes, p, bytes);
    }
    return node_barrier(n);                                                         // nobody rewrites its segment before everybody has read it
}

bool node_all_to_all_v(Node* n, const void* send, const uint64_t* send_cnt, const uint64_t* send_off, void* recv, const uint64_t* recv_cnt, const uint64_t* recv_off) {
    const int me = n->rank; uint64_t sb = 0;
    for (int r = 0; r < n->world; r++) { sb = std::max(sb, send_off[r] + send_cnt[r]); n->c->cnt[me][r] = send_cnt[r]; n->c->off[me][r] = send_off[r]; }
    if (!publish(n, send, sb)) { node_claim_failure(n); return false; }
    if (!node_barrier(n)) return false;
    for (int r = 0; r < n->world; r++) {
        if (!recv_cnt[r]) continue;
        const uint64_t o = n->c->off[r][me];
        const uint8_t* p = n->c->cnt[r][me] == recv_cnt[r] ? peer(n, r, o + recv_cnt[r]) : nullptr;   // (sizes that disagree: the protocol above is broken)
        if (!p) { node_claim_failure(n); return false; }
        memcpy((uint8_t*)recv + recv_off[r], p + o, recv_cnt[r]);
    }
    return node_barrier(n);
}

bool node_all_gather_v(Node* n, const std::string& mine, std::vector<std::string>& all) {
    std::vector<uint64_t> sizes(n->world); const uint64_t sz = mine.size();
    if (!node_all_gather(n, &sz, sizes.data(), 8)) return false;
    uint64_t mx = 0; for (uint64_t s : sizes) mx = std::max(mx, s);
    std::string pad = mine; pad.resize(mx);
    std::string got((size_t)mx * n->world, '\0');
    if (!node_all_gather(n, pad.data(), &got[0], mx)) return false;
    all.resize(n->world);
    for (int r = 0; r < n->world; r++) all[r] = got.substr((size_t)r * mx, sizes[r]);
    return true;
}

namespace {
int cb_all_gather(void* user, const void* send, void* recv, uint64_t bytes) { return node_all_gather((Node*)user, send, recv, bytes) ? 0 : 1; }
int cb_all_to_all_v(void* user, const void* send, const uint64_t* sc, const uint64_t* so, void* recv, const uint64_t* rc, const uint64_t* ro) {
    return node_all_to_all_v((Node*)user, send, sc, so, recv, rc, ro) ? 0 : 1;
}
volatile sig_atomic_t g_signal = 0;
void on_signal(int s) { g_signal = s; }
}  // namespace
skh_host_collectives node_collectives(Node* n) { return skh_host_collectives{n, cb_all_gather, cb_all_to_all_v}; }

// Forks the ranks and returns in each of them with its rank.  The launcher does not return: it waits for the ranks; the first one that ends badly (or a signal to
// the launcher) sets the failure flag, which ends every shared-memory collective; ranks that still run a few seconds later -- inside RCCL, say -- are killed.  Exit
// status: 0 when every rank returned 0, else the first failure's.
int node_launch(Node* n) {
    fflush(stdout); fflush(stderr);
    std::vector<pid_t> kid(n->world, -1);
    for (int r = 0; r < n->world; r++) {
        const pid_t p = fork();
        if (p < 0) { for (int q = 0; q < r; q++) kill(kid[q], SIGKILL); throw std::runtime_error("fork failed"); }
        if (p == 0) { n->rank = r; return r; }
        kid[r] = p;
    }
    struct sigaction sa{}; sa.sa_handler = on_signal; sigaction(SIGINT, &sa, nullptr); sigaction(SIGTERM, &sa, nullptr);
    int live = n->world, first = 0; double deadline = 0;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    while (live) {
        int st = 0; const pid_t p = waitpid(-1, &st, first || g_signal ? WNOHANG : 0);
        if (p > 0) {
            live--;
            for (int r = 0; r < n->world; r++) if (kid[r] == p) kid[r] = -1;
            const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
            if (code && !first) {
                first = code;
                if (node_claim_failure(n) && !WIFEXITED(st)) fprintf(stderr, "ERROR a rank was ended by signal %d; all ranks stop\n", WTERMSIG(st));
                deadline = now() + 5.0;
            }
            continue;
        }
        if (p < 0 && errno != EINTR && errno != ECHILD) break;
        if (p < 0 && errno == ECHILD) break;
        if (g_signal && !first) { first = 128 + g_signal; node_claim_failure(n); deadline = now() + 1.0; }
        if (first && now() > deadline) { for (int r = 0; r < n->world; r++) if (kid[r] > 0) kill(kid[r], SIGKILL); deadline = now() + 60.0; }
        if (first || g_signal) { const timespec ts{0, 20 * 1000 * 1000}; nanosleep(&ts, nullptr); }
    }
    fflush(stderr);
    _exit(first);
}

}  // namespace skhost
