// chain_estimate.h -- calculate_ani, per pair (chain.rs:414-555) + regression.rs:30-64: finalize_kernel.
// Device code of chain.hip (one translation unit: the kernels are launched by chain_pairs() there); included inside namespace skh.
#pragma once

// ------------------------------------------------------------------------------------------------ per-pair result
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct FinalizeArgs {
    uint32_t n_pairs, c, k;
    double min_af, both_min_af; int robust, median, learned, compute_ci;
    const GbdtModel::Node* nodes; const uint32_t* tree_off; uint32_t n_trees; float shrinkage, bias;
};
struct FinalizeScratch { double *u_est, *s_est; uint32_t *u_w, *s_w; uint64_t* cum; };

// fastrand 1.9.0 WyRand stream seeded with 7 (chain.rs:62); draw number d (0-based) is a pure function of d
constexpr uint64_t WYRAND_STEP = 0xA0761D6478BD642Full;
// output for generator state s: low ^ high half of the 128-bit product s * (s ^ c), from four 32x32+64 multiply-adds
__device__ __forceinline__ uint64_t wyrand_mix(uint64_t s) {
    const uint64_t b = s ^ 0xE7037ED1A0B428DBull;
    const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)s0 * b0;
    const uint64_t p01 = (uint64_t)s0 * b1 + (p00 >> 32);
    const uint64_t p10 = (uint64_t)s1 * b0 + (uint32_t)p01;
    const uint64_t p11 = (uint64_t)s1 * b1 + (p01 >> 32) + (p10 >> 32);
    return ((p10 << 32) | (uint32_t)p00) ^ p11;
}
__device__ __forceinline__ uint64_t wyrand_state(uint64_t d) { return 7ull + (d + 1ull) * WYRAND_STEP; }   // state after d + 1 steps from seed 7
__device__ __forceinline__ uint64_t wyrand_draw(uint64_t d) { return wyrand_mix(wyrand_state(d)); }

constexpr uint32_t FIN_WAVES = 2, FIN_THREADS = 64 * FIN_WAVES;   // measured on 9,500 pairs: 1 wave per pair 0.98 ms, 2: 0.67, 4: 0.70, 5: 0.97, 8: 1.04
// chain.rs:414-555 + regression.rs:30-64.  One workgroup of FIN_WAVES waves per pair.  The per-pair work arrays (one entry per chunk) live in LDS;
// the kernel is instantiated for FIN_LDS = 320 (genomes up to ~6 Mbp: 11 KB per pair) and 1024 entries and a pair runs in the smaller one that
// holds it; beyond 1024 chunks the arrays spill to global scratch.  The work is a chain of dependent LDS reads, shuffles and f64 arithmetic:
// with four pairs (one wave each) per workgroup the 45 KB of LDS held the kernel at 2.4 waves per SIMD for 1.0 ms; the rank sort and the 100 bootstrap resamples -- 9/10 of the
// instructions -- are independent per element / per resample and are dealt to the waves, the short sequential steps run on wave 0 or
// redundantly on every wave (same operations in the same order: the same bits).
template <uint32_t FIN_LDS, uint32_t FIN_MIN, uint32_t FIN_MAX, uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void finalize_kernel(FinalizeArgs fa, const PairDesc* pairs, const uint32_t* pc0, const uint32_t* n_chunks,
                                                       const double* chunk_est, const uint32_t* chunk_w, const uint4* chunk_sums, FinalizeScratch fs,
                                                       uint32_t* n_est_out, skh_ani_result* out) {
    __shared__ double lds_boot[128];
    __shared__ double lds_u[FIN_LDS], lds_s[FIN_LDS];
    __shared__ uint64_t lds_cum[FIN_LDS];
    __shared__ uint32_t lds_uw[FIN_LDS], lds_sw[FIN_LDS];
    __shared__ uint32_t lds_hdr[4];
    __shared__ uint64_t lds_total;
    const uint32_t wv = threadIdx.x >> 6, tid = threadIdx.x;
    const uint32_t p = blockIdx.x;
    if (p >= fa.n_pairs) return;
    const uint32_t l = lane_id();
    const PairDesc pd = pairs[p];
    const uint32_t C0 = pc0[p], nc = n_chunks[p];
    if (nc < FIN_MIN || nc > FIN_MAX) return;                                       // another instantiation's pair
    constexpr uint32_t WAVES = THREADS / 64;
    const bool in_lds = nc <= FIN_LDS;
    double* U = in_lds ? lds_u : fs.u_est + C0; uint32_t* UW = in_lds ? lds_uw : fs.u_w + C0;
    double* S = in_lds ? lds_s : fs.s_est + C0; uint32_t* SW = in_lds ? lds_sw : fs.s_w + C0;
    uint64_t* CUM = in_lds ? lds_cum : fs.cum + C0;
    // 1. valid (estimate, weight) pairs in chunk order (wave 0)
    if (wv == 0) {
        uint32_t n = 0, acl = 0, nchains = 0, tqb = 0;
        for (uint32_t b = 0; b < nc; b += 64) {
            const uint32_t s = C0 + b + l;
            if (b + l < nc) { const uint4 cs = chunk_sums[s]; acl += cs.x; nchains += cs.y; tqb += cs.z; }
            const bool v = b + l < nc && chunk_w[s] != NONE;
            const unsigned long long m = __ballot(v);
            if (v) { const uint32_t o = n + (uint32_t)__popcll(m & ((1ull << l) - 1ull)); U[o] = chunk_est[s]; UW[o] = chunk_w[s]; }
            n += (uint32_t)__popcll(m);
        }
        acl = wave_sum(acl); nchains = wave_sum(nchains); tqb = wave_sum(tqb);
        if (l == 0) { n_est_out[p] = n; lds_hdr[0] = n; lds_hdr[1] = acl; lds_hdr[2] = nchains; lds_hdr[3] = tqb; }
    }
    __syncthreads();
    const uint32_t n = lds_hdr[0], acl = lds_hdr[1], nchains = lds_hdr[2], tqb = lds_hdr[3];
    skh_ani_result res;
    memset(&res, 0, sizeof res);
    if (n == 0 || nchains == 0) {                                                   // chain.rs:416-420: AniEstResult::default() with ani = NaN
        res.ani = __builtin_nanf("");
        if (tid == 0) out[p] = res;
        return;
    }
    // 2. ascending sort by (estimate, weight) (chain.rs:414).  In LDS by rank counting, one element per thread.  The instantiation for pairs beyond
    //    the LDS arrays (a 2.3 Gbp pair has 115,000 chunks: counting ranks took 11 s) sorts in global memory with a bitonic network whose comparators
    //    all point the same way -- any n, no padding; elements equal in both fields are interchangeable, so no index is carried.
    if (!in_lds) {
        unsigned long long* SB = (unsigned long long*)S;                            // (estimates are non-negative: their bit patterns order like the values)
        for (uint32_t i = tid; i < n; i += THREADS) { const double d = U[i]; unsigned long long bits; memcpy(&bits, &d, 8); __atomic_store_n(&SB[i], bits, __ATOMIC_RELAXED); __atomic_store_n(&SW[i], UW[i], __ATOMIC_RELAXED); }
        block_fence(); __syncthreads();
        uint32_t N = 1; while (N < n) N <<= 1;
        auto exchange = [&](uint32_t i, uint32_t x) {                                // i < x: the smaller element to i
            if (x >= n) return;
            const unsigned long long ea = __atomic_load_n(&SB[i], __ATOMIC_RELAXED), eb = __atomic_load_n(&SB[x], __ATOMIC_RELAXED);
            if (eb > ea) return;
            const uint32_t wa = __atomic_load_n(&SW[i], __ATOMIC_RELAXED), wb = __atomic_load_n(&SW[x], __ATOMIC_RELAXED);
            if (eb == ea && wb >= wa) return;
            __atomic_store_n(&SB[i], eb, __ATOMIC_RELAXED); __atomic_store_n(&SW[i], wb, __ATOMIC_RELAXED);
            __atomic_store_n(&SB[x], ea, __ATOMIC_RELAXED); __atomic_store_n(&SW[x], wa, __ATOMIC_RELAXED);
        };
        for (uint32_t kk = 2; kk <= N; kk <<= 1) {
            const uint32_t hk = kk >> 1;
            for (uint32_t t = tid; t < N / 2; t += THREADS) { const uint32_t blk = t / hk, off = t % hk; exchange(blk * kk + off, blk * kk + (kk - 1u - off)); }
            block_fence(); __syncthreads();
            for (uint32_t j = hk >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < N / 2; t += THREADS) { const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)); exchange(i, i | j); }
                block_fence(); __syncthreads();
            }
        }
    } else
    for (uint32_t i = tid; i < n; i += THREADS) {
        const double e = U[i]; const uint32_t w = UW[i];
        uint32_t rank = 0;
#pragma unroll 4
        for (uint32_t j = 0; j < n; j++) { const double ej = U[j]; const uint32_t wj = UW[j]; rank += (ej < e || (ej == e && (wj < w || (wj == w && j < i)))) ? 1u : 0u; }
        S[rank] = e; SW[rank] = w;
    }
    __syncthreads();
    // 3. inclusive cumulative weights (wave 0)
    if (wv == 0) {
        uint64_t carry = 0;
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t i = b + l;
            uint64_t v = i < n ? SW[i] : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint64_t t = __shfl_up(v, d, 64); if (l >= (uint32_t)d) v += t; }
            if (i < n) CUM[i] = carry + v;
            carry += __shfl(v, 63, 64);
        }
        if (l == 0) lds_total = carry;
    }
    __syncthreads();
    const uint64_t total_mult = lds_total;
    // 4. quantile window (chain.rs:426-460) -- steps 4 and 5 read only: every wave computes them for itself
    double lower = 0., upper = 1.;
    if (fa.median) { lower = 0.499; upper = 0.501; } else if (fa.robust) { lower = 0.10; upper = 0.90; }
    const uint64_t thr_lo = (uint64_t)((double)total_mult * lower), thr_hi = (uint64_t)((double)total_mult * upper);
    uint32_t lower_i = n, upper_i = n;   // first indices reaching the thresholds
    for (uint32_t b = 0; b < n && (lower_i == n || upper_i == n); b += 64) {
        const uint32_t i = b + l;
        const uint64_t cv = i < n ? CUM[i] : 0;
        const unsigned long long mlo = __ballot(i < n && cv >= thr_lo), mhi = __ballot(i < n && cv >= thr_hi);
        if (lower_i == n && mlo) lower_i = b + (uint32_t)__ffsll((long long)mlo) - 1u;
        if (upper_i == n && mhi) upper_i = b + (uint32_t)__ffsll((long long)mhi) - 1u;
    }
    if (lower_i == n) lower_i = 0;
    upper_i = upper_i == n ? n - 1 : upper_i + 1;                                   // chain.rs:444,455-458
    // 5. weighted mean over [lower_i, upper_i) and population std of all estimates (chain.rs:462-471, 39-55)
    double wsum = 0., esum = 0.; uint64_t tm = 0;
    for (uint32_t i = l; i < n; i += 64) {
        const double e = S[i]; esum += e;
        if (i >= lower_i && i < upper_i) { wsum += e * (double)SW[i]; tm += SW[i]; }
    }
    wsum = wave_sum_f64(wsum); esum = wave_sum_f64(esum); tm = wave_sum_u64(tm);
    double final_ani = wsum / (double)tm;
    const double mean = esum / (double)n;
    double var = 0.;
    for (uint32_t i = l; i < n; i += 64) { const double d = mean - S[i]; var += d * d; }
    var = wave_sum_f64(var);
    const double sd = sqrt(var / (double)n);
    // 6. percentile bootstrap (chain.rs:57-86): 100 resamples of n draws from the multiplicity-expanded list, dealt to the waves in groups of four
    double ci_lo = 0., ci_hi = 1.;
    if (fa.compute_ci && n >= 10) {
        uint32_t nsteps = 0; while ((1u << nsteps) < n) nsteps++;                  // fixed-length branch-free binary search
        // first i with CUM[i] > x; x < total_mult = CUM[n-1], so the answer is in [0, n-1]
        auto search64 = [&](uint64_t x) { uint32_t lo = 0, hi = n - 1; for (uint32_t st = 0; st < nsteps; st++) { const uint32_t mid = (lo + hi) >> 1; const bool gt = CUM[mid] > x; hi = gt ? mid : hi; lo = gt ? lo : mid + 1; } return lo; };
        if (in_lds && total_mult < 0xFFFFFFFFull) {
            // Fast path (every realistic pair): 32-bit cumulative weights, and a 512-entry directory over the value range
            // (bucket b = x >> sh; entry = first | last candidate << 16, one LDS read) that narrows each search to the one or two entries
            // whose cumulative weight falls into the draw's bucket.  Both live in LDS arrays that are dead after the sort (unsorted
            // weights / estimates; 512 x 4 B fit the smaller instantiation's 320 doubles).
            uint32_t* C32 = UW; uint32_t* T = (uint32_t*)U;
            uint32_t sh = 0; while ((total_mult >> sh) >= 512) sh++;
            const uint32_t nb = (uint32_t)(total_mult >> sh) + 1;                      // x < total_mult  =>  x >> sh < nb <= 512
            for (uint32_t i = tid; i < n; i += THREADS) C32[i] = (uint32_t)CUM[i];
            for (uint32_t b = tid; b < nb; b += THREADS) T[b] = search64((uint64_t)b << sh) | (search64((uint64_t)(b + 1) << sh) << 16);   // past-the-end thresholds give n-1
            __syncthreads();
            const uint32_t tot32 = (uint32_t)total_mult;
            const uint64_t step_it = (uint64_t)n * WYRAND_STEP, step_256 = 256ull * WYRAND_STEP;
            // resamples in groups of four: the four wave reductions (six dependent shuffle steps each) then run interleaved; group q -> wave q mod WAVES
            for (uint32_t it0 = 4 * wv; it0 < 100; it0 += 4 * WAVES) {
              // generator states of this lane's draws j = l + 64 u (+ 256 m) of resample `it0`: a pure function of the draw number it0 * n + j;
              // advanced by n steps per resample instead of being recomputed (a 64-bit multiply per draw)
              uint64_t st[4];
#pragma unroll
              for (int u = 0; u < 4; u++) st[u] = wyrand_state((uint64_t)it0 * n + (uint64_t)l + 64u * (uint32_t)u);
              double sg[4];
#pragma unroll
              for (int g = 0; g < 4; g++) {
                double s = 0.;
                uint64_t sm[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { sm[u] = st[u]; st[u] += step_it; }
                for (uint32_t j0 = l; j0 < n; j0 += 256) {
                    uint32_t x[4], lo[4], hi[4]; bool on[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t j = j0 + 64u * (uint32_t)u;
                        on[u] = j < n;
                        const uint64_t r = wyrand_mix(sm[u]); sm[u] += step_256;
                        // Lemire reduction hi64(r * total) for total < 2^32; its rejection branch has probability total/2^64
                        x[u] = (uint32_t)(((uint64_t)(uint32_t)(r >> 32) * tot32 + __umulhi((uint32_t)r, tot32)) >> 32);
                        const uint32_t tb = T[x[u] >> sh];
                        lo[u] = tb & 0xFFFFu; hi[u] = tb >> 16;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        while (lo[u] < hi[u]) { const uint32_t mid = (lo[u] + hi[u]) >> 1; const bool gt = C32[mid] > x[u]; hi[u] = gt ? mid : hi[u]; lo[u] = gt ? lo[u] : mid + 1; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) if (on[u]) s += S[lo[u]];
                }
                sg[g] = s;
              }
#pragma unroll
              for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
                  for (int g = 0; g < 4; g++) sg[g] += __shfl_xor(sg[g], d, 64);
              }
              if (l == 0) {
#pragma unroll
                  for (int g = 0; g < 4; g++) lds_boot[it0 + g] = sg[g] / (double)n;
              }
            }
        } else {
            for (uint32_t it = wv; it < 100; it += WAVES) {
                double s = 0.;
                for (uint32_t j0 = l; j0 < n; j0 += 64) {
                    const uint64_t r = wyrand_draw((uint64_t)it * n + j0);
                    s += S[search64(__umul64hi(r, total_mult))];
                }
                s = wave_sum_f64(s);
                if (l == 0) lds_boot[it] = s / (double)n;
            }
        }
        __syncthreads();
        if (tid < 100) {
            const uint32_t i = tid;
            const double e = lds_boot[i]; uint32_t rank = 0;
            for (uint32_t j = 0; j < 100; j++) { const double ej = lds_boot[j]; rank += (ej < e || (ej == e && j < i)) ? 1u : 0u; }
            if (rank == 4) lds_boot[100] = e;
            if (rank == 94) lds_boot[101] = e;
        }
        __syncthreads();
        ci_lo = lds_boot[100]; ci_hi = lds_boot[101];
    }
    // 7. aligned fractions, cut-offs, output record (chain.rs:477-554) -- computed redundantly by every lane (wave-uniform)
    double cov_q = (double)tqb / (double)pd.query_total_len; if (!(cov_q < 1.)) cov_q = 1.;
    double cov_r = (double)tqb / (double)pd.ref_total_len; if (!(cov_r < 1.)) cov_r = 1.;   // total_ref_range has the same numerator (chain.rs:245-246)
    const double cutoff = fa.min_af < 0. ? 0.15 : fa.min_af;                        // chain.rs:100-107
    if (fa.both_min_af > 0.0) { if (cov_q < fa.both_min_af || cov_r < fa.both_min_af) final_ani = -1.; }
    else if (cov_q < cutoff && cov_r < cutoff) final_ani = -1.;
    res.ani = (float)final_ani; res.af_query = (float)cov_q; res.af_ref = (float)cov_r;
    res.ci_lower = (float)ci_lo; res.ci_upper = (float)ci_hi; res.std = (float)sd;
    res.q90_q = pd.q90_q; res.q90_r = pd.q90_r; res.q50_q = pd.q50_q; res.q50_r = pd.q50_r; res.q10_q = pd.q10_q; res.q10_r = pd.q10_r;
    res.num_contigs_q = pd.nctg_q; res.num_contigs_r = pd.nctg_r;
    res.avg_chain_int_len = acl / nchains;                                          // chain.rs:421
    res.total_bases_covered = tqb;
    // 8. learned ANI (regression.rs:30-64; gbdt 0.1.1 LAD predict = bias + sum shrinkage * leaf, f32, tree order).
    //    The 195 tree walks are independent: one tree per thread; the f32 sum stays sequential in tree order.
    if (fa.learned && res.ani > 0.9f && res.total_bases_covered > REGRESS_CUTOFF) { // wave-uniform condition
        float x[5];
        x[0] = res.ani * 100.f; x[1] = res.std; x[4] = (float)res.avg_chain_int_len;
        if (res.q50_r > res.q50_q) { x[2] = res.q90_r; x[3] = res.q90_q; } else { x[2] = res.q90_q; x[3] = res.q90_r; }
        float* leaf = (float*)lds_boot;                                             // 256 floats
        __syncthreads();                                                            // (every thread has read its interval bounds from lds_boot)
        for (uint32_t t = tid; t < fa.n_trees && t < 256; t += THREADS) {
            const GbdtModel::Node* nd = fa.nodes + fa.tree_off[t]; int32_t i = 0;
            while (nd[i].feat >= 0) {
                const int32_t ft = nd[i].feat;
                const float xv = ft == 0 ? x[0] : ft == 1 ? x[1] : ft == 2 ? x[2] : ft == 3 ? x[3] : x[4];
                i = xv < nd[i].thr ? nd[i].left : nd[i].right;
            }
            leaf[t] = nd[i].pred;
        }
        __syncthreads();
        if (tid == 0) {
            float pred = fa.bias;
            for (uint32_t t = 0; t < fa.n_trees; t++) {
                float lv;
                if (t < 256) lv = leaf[t];
                else { const GbdtModel::Node* nd = fa.nodes + fa.tree_off[t]; int32_t i = 0; while (nd[i].feat >= 0) i = x[nd[i].feat] < nd[i].thr ? nd[i].left : nd[i].right; lv = nd[i].pred; }
                pred += fa.shrinkage * lv;
            }
            if (pred < 100.f) {
                res.ci_upper = (res.ci_upper - res.ani) + pred / 100.f;
                res.ci_lower = (res.ci_lower - res.ani) + pred / 100.f;
                res.ani = pred / 100.f;
            }
        }
    }
    if (tid == 0) out[p] = res;
}
