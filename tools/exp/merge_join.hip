// merge_join.hip -- round 5, VERDICT r04 item 5: what would a STORE-side join cost?  Measured before it is built (the way tools/exp/virtual_join.py did for
// the per-cluster index), on synthetic sketches with the headline workload's statistics: 1000 genomes in clades of 20, ~38,000 distinct seeds each, a member
// keeps a seed of its clade's root with probability 0.45 .. 0.93 (ANI^15 for 0.5-8 % divergence), 9,500 pairs.
//
// The formulation: every genome keeps its distinct seeds in HASH ORDER (keys), with a payload word (position / list head) and the seed's position INDEX in the
// genome's position-ordered arrays, cut into 1024 buckets by the hash's leading 10 bits.  A workgroup takes BG consecutive buckets of one pair: both sides'
// stretches go through LDS (coalesced streams), every A entry looks its hash up in B's stretch (LDS only), and a common seed's (payload A, payload B) record is
// STORED at hit[pair][index of A's position] -- independent scattered stores, no dependent gather.  (The existing fill pass would then read hit[] in position
// order and emit the anchors already sorted, chain.rs:721.)
// What the real kernel would add on top: the band rules (chain.rs:674-676, 694-696), list heads, the per-tile anchor counts, the in-query mask.  This is the floor.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/merge_join tools/exp/merge_join.hip      run: tools/exp/merge_join [n_genomes] [seeds_per_genome]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t NB = 1024;                                                       // buckets per genome (leading 10 bits of the 32-bit hash)

struct Pair { uint32_t a, b; uint64_t hit0; };                                       // genome numbers; first record of the pair's hit array

// LOOKUP 0: binary search in B's stretch; 1: interpolation (hashes are uniform) + a short walk
template <uint32_t BG, int LOOKUP>
__global__ __launch_bounds__(256) void merge_kernel(const Pair* __restrict__ pairs, const uint32_t* __restrict__ queue /* [8][per_q] pair numbers, 0xFFFFFFFF = none */, uint32_t per_q,
                                                    const uint64_t* __restrict__ g_off /* first entry of every genome */, const uint32_t* __restrict__ b_off /* [genome][NB + 1] */,
                                                    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pay, const uint32_t* __restrict__ idx,
                                                    uint2* __restrict__ hit, uint32_t* __restrict__ n_hits) {
    constexpr uint32_t CAP = BG * 48;                                               // entries a stretch may hold (mean 37 per bucket, Poisson: 7 sigma at BG = 16)
    __shared__ uint32_t ka[CAP], pa[CAP], ia[CAP], kb[CAP], pb[CAP];
    __shared__ uint32_t cnt;
    const uint32_t groups = NB / BG;
    const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
    const uint32_t pq = queue[(uint64_t)xcd * per_q + local / groups];
    if (pq == 0xFFFFFFFFu) return;
    const uint32_t grp = local % groups;
    const Pair P = pairs[pq];
    const uint32_t a0 = b_off[(uint64_t)P.a * (NB + 1) + grp * BG], a1 = b_off[(uint64_t)P.a * (NB + 1) + (grp + 1) * BG];
    const uint32_t b0 = b_off[(uint64_t)P.b * (NB + 1) + grp * BG], b1 = b_off[(uint64_t)P.b * (NB + 1) + (grp + 1) * BG];
    const uint32_t na = min(a1 - a0, CAP), nb = min(b1 - b0, CAP);
    const uint64_t oa = g_off[P.a] + a0, ob = g_off[P.b] + b0;
    if (threadIdx.x == 0) cnt = 0;
    for (uint32_t i = threadIdx.x; i < na; i += 256) { ka[i] = keys[oa + i]; pa[i] = pay[oa + i]; ia[i] = idx[oa + i]; }
    for (uint32_t i = threadIdx.x; i < nb; i += 256) { kb[i] = keys[ob + i]; pb[i] = pay[ob + i]; }
    __syncthreads();
    uint32_t mine = 0;
    const uint32_t lo_key = (grp * BG) << 22, span_log = 32 - 10 + __builtin_ctz(BG);   // the stretch covers hashes [lo_key, lo_key + 2^span_log)
    for (uint32_t i = threadIdx.x; i < na; i += 256) {
        const uint32_t k = ka[i];
        uint32_t at;
        if (LOOKUP == 0) {
            uint32_t lo = 0, hi = nb;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (kb[mid] < k) lo = mid + 1; else hi = mid; }
            at = lo;
        } else {
            at = (uint32_t)(((uint64_t)(k - lo_key) * nb) >> span_log);
            if (at >= nb) at = nb ? nb - 1 : 0;
            while (at > 0 && kb[at] > k) at--;
            while (at < nb && kb[at] < k) at++;
        }
        if (at < nb && kb[at] == k) { hit[P.hit0 + ia[i]] = make_uint2(pa[i], pb[at]); mine++; }
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(&n_hits[pq], cnt);
}

int main(int argc, char** argv) {
    const uint32_t NG = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000, NS = argc > 2 ? (uint32_t)atoi(argv[2]) : 38000, CL = 20;
    std::mt19937_64 rng(12345);
    std::vector<std::vector<uint32_t>> sets(NG);
    for (uint32_t c = 0; c < NG / CL; c++) {
        std::vector<uint32_t> root(NS); for (auto& x : root) x = (uint32_t)rng();
        for (uint32_t m = 0; m < CL; m++) {
            const double keep = 0.45 + 0.48 * (double)(rng() % 1000) / 1000.0;
            auto& s = sets[c * CL + m]; s.resize(NS);
            for (uint32_t i = 0; i < NS; i++) s[i] = ((double)(rng() % 100000) / 100000.0 < keep) ? root[i] : (uint32_t)rng();
            std::sort(s.begin(), s.end()); s.erase(std::unique(s.begin(), s.end()), s.end());
        }
    }
    std::vector<uint64_t> g_off(NG + 1, 0);
    for (uint32_t g = 0; g < NG; g++) g_off[g + 1] = g_off[g] + sets[g].size();
    const uint64_t TOT = g_off[NG];
    std::vector<uint32_t> keys(TOT), pay(TOT), idx(TOT), b_off((size_t)NG * (NB + 1));
    for (uint32_t g = 0; g < NG; g++) {
        const auto& s = sets[g]; const uint32_t n = (uint32_t)s.size();
        std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u); std::shuffle(perm.begin(), perm.end(), rng);   // hash order has nothing to do with position order
        for (uint32_t i = 0; i < n; i++) { keys[g_off[g] + i] = s[i]; pay[g_off[g] + i] = (uint32_t)rng() >> 1; idx[g_off[g] + i] = perm[i]; }
        for (uint32_t b = 0; b < NB; b++) b_off[(size_t)g * (NB + 1) + b] = (uint32_t)(std::lower_bound(s.begin(), s.end(), b << 22) - s.begin());
        b_off[(size_t)g * (NB + 1) + NB] = n;
    }
    std::vector<Pair> pairs; uint64_t hit0 = 0;
    for (uint32_t c = 0; c < NG / CL; c++) for (uint32_t i = 0; i < CL; i++) for (uint32_t j = i + 1; j < CL; j++) { pairs.push_back({c * CL + j, c * CL + i, hit0}); hit0 += sets[c * CL + j].size(); }
    const uint32_t NP = (uint32_t)pairs.size();
    std::vector<std::vector<uint32_t>> q(8);
    for (uint32_t p = 0; p < NP; p++) q[pairs[p].b & 7u].push_back(p);                // pairs that probe one sketch share an XCD, one after the other
    uint32_t per_q = 0; for (auto& v : q) per_q = std::max<uint32_t>(per_q, (uint32_t)v.size());
    std::vector<uint32_t> queue((size_t)8 * per_q, 0xFFFFFFFFu);
    for (uint32_t x = 0; x < 8; x++) std::copy(q[x].begin(), q[x].end(), queue.begin() + (size_t)x * per_q);
    uint32_t *d_keys, *d_pay, *d_idx, *d_boff, *d_queue, *d_nh; uint64_t* d_goff; Pair* d_pairs; uint2* d_hit;
    CK(hipMalloc(&d_keys, TOT * 4)); CK(hipMalloc(&d_pay, TOT * 4)); CK(hipMalloc(&d_idx, TOT * 4)); CK(hipMalloc(&d_boff, b_off.size() * 4)); CK(hipMalloc(&d_queue, queue.size() * 4));
    CK(hipMalloc(&d_nh, NP * 4)); CK(hipMalloc(&d_goff, (NG + 1) * 8)); CK(hipMalloc(&d_pairs, NP * sizeof(Pair))); CK(hipMalloc(&d_hit, hit0 * 8));
    CK(hipMemcpy(d_keys, keys.data(), TOT * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pay, pay.data(), TOT * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_idx, idx.data(), TOT * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_boff, b_off.data(), b_off.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_queue, queue.data(), queue.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_goff, g_off.data(), (NG + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pairs, pairs.data(), NP * sizeof(Pair), hipMemcpyHostToDevice));
    printf("%u genomes, %.0f distinct seeds each, %u pairs; per pair: A 12 B + B 8 B per seed streamed = %.2f MB, %.1f GB in all; hit array %.2f GB\n", NG, (double)TOT / NG, NP,
           20.0 * TOT / NG / 1e6, 20.0 * TOT / NG * NP / 1e9, hit0 * 8 / 1e9);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kernel, uint32_t bg) {
        const uint32_t groups = NB / bg; const unsigned grid = per_q * groups * 8;
        float best = 1e9f; uint64_t hits = 0;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(d_nh, 0, NP * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, (const Pair*)d_pairs, (const uint32_t*)d_queue, per_q, (const uint64_t*)d_goff, (const uint32_t*)d_boff,
                               (const uint32_t*)d_keys, (const uint32_t*)d_pay, (const uint32_t*)d_idx, d_hit, d_nh);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        std::vector<uint32_t> nh(NP); CK(hipMemcpy(nh.data(), d_nh, NP * 4, hipMemcpyDeviceToHost)); for (uint32_t v : nh) hits += v;
        printf("%-44s %8.3f ms  (%u workgroups, %.1f M common seeds = %.0f per pair)\n", name, best, grid, hits / 1e6, (double)hits / NP);
    };
    run("BG 16, binary search", merge_kernel<16, 0>, 16);
    run("BG 32, binary search", merge_kernel<32, 0>, 32);
    run("BG 64, binary search", merge_kernel<64, 0>, 64);
    run("BG 16, interpolation", merge_kernel<16, 1>, 16);
    run("BG 32, interpolation", merge_kernel<32, 1>, 32);
    run("BG 64, interpolation", merge_kernel<64, 1>, 64);
    return 0;
}
