// screen_keys.hip -- the screen's incidence list: the (marker, genome) keys of marker sets (common.h screen_key), sorted by the marker's leading 16 bases.
//
// What screen.rs:190-210 (kmer_to_sketch_from_refs) builds as a hash map from marker to genome ids is, on the device, one array of keys in which the incidences of a marker
// sit next to each other (screen.hip).  Rounds 1-4 made it with a device-wide LSD radix sort (rocPRIM: a histogram kernel, four passes, eight fills; its workgroups of 512
// threads with 32 KB of LDS find no room beside the table build, whose 256-thread workgroups refill every slot that frees up -- the histogram kernel took 0.9 ms there,
// 25 us alone -- and the screen waited 0.18 ms for the sort's tail).  This file is the sort the input allows instead:
//   * every genome's marker set is SORTED, so the markers of a range of values are a stretch of every set, found by two binary searches;
//   * the order inside a group of equal prefixes is free (the count kernels compare whole markers), so nothing has to be stable.
// Three steps, workgroups of 256 threads, no pass over the keys that does not have to be:
//   1. tiles of (a range of <= 1024 buckets) x (a group of genomes): the stretches' markers are counted per bucket in LDS, one global add per bucket and tile (hist);
//      a one-workgroup scan turns the counts into bucket offsets; the same tiles again, now asking a global cursor for each bucket's share of the tile and
//      writing the keys there (scatter).  Buckets are equal slices of t = 2x - x^2 (x the prefix as a fraction of 2^32): a marker is the smaller of a 21-mer and its reverse
//      complement, so x has density 2(1 - x) and t is uniform -- with equal slices of x the first bucket would hold twice the average.
//   2. a workgroup per bucket (~1,400 keys, at most SKEYS_CAP_MAX in LDS): the next 8 bits of t split it into 256 fine groups with LDS atomics, and a key's rank inside
//      its fine group is counted (a handful of keys; a fine group of ONE prefix -- a marker shared by thousands of genomes -- needs no ranks at all).
// A bucket beyond the LDS capacity (tens of thousands of genomes sharing their markers) sends the bucketed keys through the radix sort after all; the decision is made on
// the host from the scan's maximum, at a point where the caller synchronises anyway.
#include <algorithm>

#include "internal.h"

namespace skh {

namespace {

constexpr uint32_t SKEYS_T = 256;
constexpr uint32_t SKEYS_RB_MAX = 1024;               // buckets a tile counts in LDS
constexpr uint32_t SKEYS_GG_MAX = 256;                // genomes of a tile (one thread each looks up the genome's stretch)
constexpr uint32_t SKEYS_CAP_MAX = 8192;              // keys of a bucket the second step holds in LDS (12 B per key)
constexpr uint32_t SKEYS_NB_MAX = 1u << 20;

struct BucketMap { uint32_t t_base, shift, nb; };   // (what the kernels take of a ScreenKeysPlan)

// x -> 2x - x^2 on 32-bit fractions, non-decreasing: (x + 1)^2 - x^2 < 2^33, so the subtracted floor grows by at most 2 per step of x
__host__ __device__ __forceinline__ uint32_t skeys_t(uint32_t prefix) {
    const uint64_t x = prefix, t = 2 * x - ((x * x) >> 32);
    return t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
}
__device__ __forceinline__ uint32_t skeys_rel(const BucketMap& bm, uint32_t prefix) { const uint32_t t = skeys_t(prefix); return t >= bm.t_base ? t - bm.t_base : 0u; }
__device__ __forceinline__ uint32_t skeys_bucket(const BucketMap& bm, uint64_t marker) {
    const uint32_t b = (uint32_t)((uint64_t)skeys_rel(bm, (uint32_t)(marker >> 10)) >> bm.shift);
    return b < bm.nb ? b : bm.nb - 1;                                                // (a marker of the stated range never needs the clamp)
}

// where, inside genome g's stretch, the markers of bucket range r begin: bounds[g * (n_ranges + 1) + r]; entry n_ranges is the stretch's length.  A thread per entry: the
// binary searches of all tiles at once (made by the tiles themselves, each tile began with two dozen dependent loads per genome, twice).
__global__ __launch_bounds__(SKEYS_T) void skeys_bounds_kernel(ScreenKeysIn in, BucketMap bm, uint32_t rb, uint32_t n_ranges, uint32_t* bounds) {
    const uint64_t t = (uint64_t)blockIdx.x * SKEYS_T + threadIdx.x; const uint32_t per = n_ranges + 1;
    if (t >= (uint64_t)in.ng * per) return;
    const uint32_t g = (uint32_t)(t / per), r = (uint32_t)(t % per);
    const uint64_t a = in.range_lo ? in.range_lo[g] : in.mk_off[g];
    const uint64_t e = a + (in.range_cnt ? (uint64_t)in.range_cnt[g] : in.mk_off[g + 1] - in.mk_off[g]);
    uint64_t lo = a, hi = e;
    if (r == 0) hi = a;
    else if (r == n_ranges) lo = e;
    else { const uint32_t v = r * rb; while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (skeys_bucket(bm, in.markers[mid]) < v) lo = mid + 1; else hi = mid; } }
    bounds[t] = (uint32_t)(lo - a);
}

// A tile: the markers of bucket range blockIdx.x in the stretches of genomes [g0, g0 + gg).  Its keys are walked as ONE list (a key's genome by a search in the LDS prefix of
// the stretch lengths): with a loop per genome, stretches shorter than the workgroup -- 10,000 genomes, 150 keys each -- left most threads idle behind one load at a time.
template <bool SCATTER>
__global__ __launch_bounds__(SKEYS_T) void skeys_tile_kernel(ScreenKeysIn in, BucketMap bm, uint32_t rb, uint32_t gg, uint32_t n_ranges, const uint32_t* __restrict__ bounds,
                                                             uint32_t* counters /* hist | cursors */, uint64_t* out) {
    __shared__ uint32_t cnt[SKEYS_RB_MAX];
    __shared__ uint32_t base[SCATTER ? SKEYS_RB_MAX : 1];
    __shared__ uint64_t sx[SKEYS_GG_MAX];
    __shared__ uint32_t sp[SKEYS_GG_MAX + 1];
    __shared__ uint32_t wsum[SKEYS_T / 64];
    const uint32_t tid = threadIdx.x, b0 = blockIdx.x * rb, g0 = blockIdx.y * gg, per = n_ranges + 1;
    for (uint32_t b = tid; b < rb; b += SKEYS_T) cnt[b] = 0;
    {
        const uint32_t g = g0 + tid;
        uint64_t x = 0; uint32_t n = 0;
        if (tid < gg && g < in.ng) {
            const uint64_t a = in.range_lo ? in.range_lo[g] : in.mk_off[g];
            const uint32_t lo = bounds[(uint64_t)g * per + blockIdx.x], hi = bounds[(uint64_t)g * per + blockIdx.x + 1];
            x = a + lo; n = hi - lo;
        }
        sx[tid] = x;
        const uint32_t incl = wave_incl_scan(n), l = tid & 63u, w = tid >> 6;
        if (l == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t q = 0; q < w; q++) before += wsum[q];
        sp[tid] = before + incl - n;
        if (tid == SKEYS_T - 1) sp[SKEYS_T] = before + incl;
    }
    __syncthreads();
    const uint32_t total = sp[gg];
    constexpr uint32_t U = 4;                                                        // loads in flight per thread
    auto genome_of = [&](uint32_t i) { uint32_t lo = 0, hi = gg; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sp[mid] <= i) lo = mid; else hi = mid; } return lo; };
    for (uint32_t i0 = tid; i0 < total; i0 += U * SKEYS_T) {
        uint64_t m[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = i0 + u * SKEYS_T;
            m[u] = 0ull;
            if (i < total) { const uint32_t gi = genome_of(i); m[u] = in.markers[sx[gi] + (i - sp[gi])]; }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t b = skeys_bucket(bm, m[u]) - b0;
            if (i0 + u * SKEYS_T < total && b < rb) atomicAdd(&cnt[b], 1u);
        }
    }
    __syncthreads();
    if (!SCATTER) {
        for (uint32_t b = tid; b < rb; b += SKEYS_T) if (cnt[b]) atomicAdd(&counters[b0 + b], cnt[b]);
        return;
    }
    for (uint32_t b = tid; b < rb; b += SKEYS_T) { const uint32_t c = cnt[b]; base[SCATTER ? b : 0] = c ? atomicAdd(&counters[b0 + b], c) : 0u; cnt[b] = 0; }
    __syncthreads();
    for (uint32_t i0 = tid; i0 < total; i0 += U * SKEYS_T) {
        uint64_t m[U]; uint32_t gi[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = i0 + u * SKEYS_T;
            m[u] = 0ull; gi[u] = 0;
            if (i < total) { gi[u] = genome_of(i); m[u] = in.markers[sx[gi[u]] + (i - sp[gi[u]])]; }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t b = skeys_bucket(bm, m[u]) - b0;
            if (i0 + u * SKEYS_T < total && b < rb) out[base[SCATTER ? b : 0] + atomicAdd(&cnt[b], 1u)] = screen_key(m[u], in.is_query, g0 + gi[u]);
        }
    }
}

// bucket counts -> offsets (off[nb] = the total), the scatter's cursors, the largest count.  One workgroup: a few thousand counters.
__global__ __launch_bounds__(1024) void skeys_scan_kernel(const uint32_t* hist, uint32_t nb, uint32_t* off, uint32_t* cursor, uint32_t* max_out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t wmax;
    const uint32_t tid = threadIdx.x, l = tid & 63u, w = tid >> 6, per = (nb + 1023u) / 1024u;
    const uint32_t b0 = tid * per, b1 = b0 + per < nb ? b0 + per : nb;
    if (tid == 0) wmax = 0;
    uint32_t sum = 0, mx = 0;
    for (uint32_t b = b0; b < b1; b++) { const uint32_t c = hist[b]; sum += c; mx = c > mx ? c : mx; }
    const uint32_t incl = wave_incl_scan(sum);
    if (l == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t run = incl - sum, tot = 0;
    for (uint32_t q = 0; q < 16; q++) { const uint32_t t = wsum[q]; if (q < w) run += t; tot += t; }
    atomicMax(&wmax, mx);
    for (uint32_t b = b0; b < b1; b++) { off[b] = run; cursor[b] = run; run += hist[b]; }
    __syncthreads();
    if (tid == 0) { off[nb] = tot; *max_out = wmax; }
}

__global__ __launch_bounds__(SKEYS_T) void skeys_bucket_sort_kernel(const uint64_t* bucketed, const uint32_t* off, BucketMap bm, uint32_t cap, uint64_t* out) {
    SKH_DYN_SMEM(smem);
    uint64_t* k = (uint64_t*)smem;                                                   // the bucket's keys as they arrived
    uint16_t* rnk = (uint16_t*)(k + cap);                                            // a key's arrival number inside its fine group
    uint16_t* grp = rnk + cap;                                                       // the keys' indices, fine group by fine group
    __shared__ uint32_t fcnt[256], fmin[256], fmax[256], foff[257];
    __shared__ uint32_t wsum[SKEYS_T / 64];
    const uint32_t tid = threadIdx.x, s0 = off[blockIdx.x], n = off[blockIdx.x + 1] - s0;
    if (n == 0 || n > cap) return;                                                   // (n > cap: the host has looked at the maximum and does not launch this kernel then)
    const uint32_t fbits = bm.shift < 8u ? bm.shift : 8u, fshift = bm.shift - fbits, fmask = (1u << fbits) - 1u;
    auto fine = [&](uint32_t prefix) { return (skeys_rel(bm, prefix) >> fshift) & fmask; };
    fcnt[tid] = 0; fmin[tid] = 0xFFFFFFFFu; fmax[tid] = 0;
    __syncthreads();
    for (uint32_t i0 = tid; i0 < n; i0 += 4 * SKEYS_T) {                               // (four loads in flight)
        uint64_t key[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) { const uint32_t i = i0 + u * SKEYS_T; key[u] = i < n ? bucketed[s0 + i] : 0ull; }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * SKEYS_T;
            if (i >= n) break;
            const uint32_t prefix = (uint32_t)key[u], f = fine(prefix);
            k[i] = key[u];
            rnk[i] = (uint16_t)atomicAdd(&fcnt[f], 1u);
            atomicMin(&fmin[f], prefix); atomicMax(&fmax[f], prefix);
        }
    }
    __syncthreads();
    {
        const uint32_t v = fcnt[tid], incl = wave_incl_scan(v), l = tid & 63u, w = tid >> 6;
        if (l == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t q = 0; q < w; q++) before += wsum[q];
        foff[tid] = before + incl - v;
        if (tid == SKEYS_T - 1) foff[256] = before + incl;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += SKEYS_T) grp[foff[fine((uint32_t)k[i])] + rnk[i]] = (uint16_t)i;
    __syncthreads();
    for (uint32_t p = tid; p < n; p += SKEYS_T) {
        const uint64_t key = k[grp[p]];
        const uint32_t prefix = (uint32_t)key, f = fine(prefix), s = foff[f], e = foff[f + 1];
        uint32_t r = p - s;                                                          // one prefix in the whole group: any order will do
        if (fmin[f] != fmax[f]) {
            r = 0;
            for (uint32_t q = s; q < e; q++) { const uint32_t o = (uint32_t)k[grp[q]]; r += (o < prefix || (o == prefix && q < p)) ? 1u : 0u; }
        }
        out[s0 + s + r] = key;
    }
}

uint32_t pow2_ceil(uint64_t v) { uint32_t p = 1; while (p < v && p < (1u << 30)) p <<= 1; return p; }

}  // namespace

bool sorted_screen_keys_fits(uint64_t n) { return n < 0xFFFFFFF0ull; }

// Step 1 up to the scan: the buckets are laid out for ~n_planned keys (the exact number may come later), counted and turned into offsets; the largest count lands in *d_max
// (device memory of the caller: it travels to the host with whatever the caller reads back next).
void screen_keys_count(skh_ctx* ctx, const ScreenKeysIn& in, uint64_t n_planned, uint64_t marker_lo, uint64_t marker_hi, PendingSort* own, uint32_t* d_max, ScreenKeysPlan& pl) {
    pl = ScreenKeysPlan{};
    if (n_planned == 0 || in.ng == 0) return;
    if (!sorted_screen_keys_fits(n_planned)) throw Error("sorted_screen_keys: more than 2^32 incidences in one list");
    // the buckets: equal slices of t over the stated range of markers, ~skeys_avg keys each
    const uint32_t t_base = skeys_t((uint32_t)(marker_lo >> 10));
    const uint64_t t_last = marker_hi ? skeys_t((uint32_t)(marker_hi >> 10)) : 0xFFFFFFFFull, span = t_last > t_base ? t_last - t_base : 0;
    pl.radix_only = ctx->tune.screen_sort_radix != 0;                                  // the form before round 5: one bucket, the radix sort over all of it
    const uint32_t avg = std::max<uint32_t>(ctx->tune.skeys_avg, 16u);
    const uint32_t want = pl.radix_only ? 1u : std::min<uint32_t>(pow2_ceil((n_planned + avg - 1) / avg), SKEYS_NB_MAX);
    pl.t_base = t_base; pl.shift = 0;
    while ((span >> pl.shift) + 1 > want) pl.shift++;
    pl.nb = (uint32_t)((span >> pl.shift) + 1);
    // a tile: ~8,192 keys -- (the buckets of a range) x (the genomes of a group).  Many small genomes: 1,024 buckets, several genomes; a few large ones: fewer buckets, one genome
    const double per_genome_bucket = (double)n_planned / (double)in.ng / (double)pl.nb;
    pl.rb = std::min<uint32_t>(pow2_ceil(pl.nb), SKEYS_RB_MAX);
    while (pl.rb > 16 && per_genome_bucket * pl.rb > 8192.) pl.rb >>= 1;
    pl.n_ranges = (pl.nb + pl.rb - 1) / pl.rb; pl.nbp = pl.n_ranges * pl.rb;
    const uint64_t per_genome_range = std::max<uint64_t>(1, (uint64_t)(per_genome_bucket * pl.rb));
    pl.gg = (uint32_t)std::min<uint64_t>(pow2_ceil(std::max<uint64_t>(1, 8192 / per_genome_range)), SKEYS_GG_MAX);
    while ((in.ng + pl.gg - 1) / pl.gg > 65535u) pl.gg <<= 1;
    if (pl.gg > SKEYS_GG_MAX) throw Error("sorted_screen_keys: too many genomes");
    pl.n_groups = (in.ng + pl.gg - 1) / pl.gg;
    // scratch: hist[nbp] off[nbp + 1] cursor[nbp], the genomes' range bounds.  A sort that outlives the call keeps them in its own buffer.
    const size_t n_bounds = (size_t)in.ng * (pl.n_ranges + 1), n_words = (size_t)3 * pl.nbp + 8 + n_bounds;
    uint32_t* words;
    if (own) { own->tmp.alloc(n_words * 4); words = (uint32_t*)own->tmp.p; } else words = ctx->arena.get<uint32_t>(n_words);
    pl.hist = words; pl.off = pl.hist + pl.nbp; pl.cursor = pl.off + pl.nbp + 1; pl.bounds = pl.cursor + pl.nbp + 4;
    const BucketMap bm{pl.t_base, pl.shift, pl.nb};
    dzero(pl.hist, (size_t)pl.nbp * 4, ctx->stream);
    SKH_LAUNCH(skeys_bounds_kernel, (unsigned)((n_bounds + SKEYS_T - 1) / SKEYS_T), SKEYS_T, 0, ctx->stream, in, bm, pl.rb, pl.n_ranges, pl.bounds);
    check_launch("skeys_bounds");
    SKH_LAUNCH(skeys_tile_kernel<false>, dim3(pl.n_ranges, pl.n_groups), SKEYS_T, 0, ctx->stream, in, bm, pl.rb, pl.gg, pl.n_ranges, (const uint32_t*)pl.bounds, pl.hist, (uint64_t*)nullptr);
    check_launch("skeys_hist");
    SKH_LAUNCH(skeys_scan_kernel, 1u, 1024, 0, ctx->stream, (const uint32_t*)pl.hist, pl.nb, pl.off, pl.cursor, d_max);
    check_launch("skeys_scan");
    pl.valid = true;
}

// The rest: the keys into their buckets, every bucket ordered into `out` -- or, when the largest bucket (h_max, read back by the caller) does not fit the LDS, through the
// radix sort.  `in` names the same marker stretches as the count did (they may have moved: the marker build counts them where they were deduplicated).
void screen_keys_place(skh_ctx* ctx, const ScreenKeysIn& in, uint64_t n, const ScreenKeysPlan& pl, uint32_t h_max, uint64_t* out, PendingSort* own) {
    if (n == 0 || in.ng == 0) return;
    if (!pl.valid || !sorted_screen_keys_fits(n)) throw Error("screen_keys_place: no plan");
    StageTrace tr(ctx);
    uint64_t* bucketed;
    if (own) { own->raw.alloc(n); bucketed = own->raw.p; } else bucketed = ctx->arena.get<uint64_t>(n);
    const BucketMap bm{pl.t_base, pl.shift, pl.nb};
    SKH_LAUNCH(skeys_tile_kernel<true>, dim3(pl.n_ranges, pl.n_groups), SKEYS_T, 0, ctx->stream, in, bm, pl.rb, pl.gg, pl.n_ranges, (const uint32_t*)pl.bounds, pl.cursor, bucketed);
    check_launch("skeys_scatter");
    const uint32_t cap_max = std::min<uint32_t>(ctx->tune.skeys_cap ? ctx->tune.skeys_cap : SKEYS_CAP_MAX, SKEYS_CAP_MAX);
    if (pl.radix_only || h_max > cap_max) {
        if (own) dsync(ctx->stream);                                                 // (the scatter reads its cursors out of the buffer the radix sort is about to take)
        sort_keys_u64_into(ctx, bucketed, out, n, 32, own ? &own->tmp : nullptr);
        tr.mark("screen keys: radix sort (a bucket beyond the LDS)");
        return;
    }
    const uint32_t cap = std::max<uint32_t>((h_max + 255u) & ~255u, 256u);
    const size_t lds = (size_t)cap * 12;
    kernel_allow_lds(skeys_bucket_sort_kernel, lds);
    SKH_LAUNCH(skeys_bucket_sort_kernel, pl.nb, SKEYS_T, lds, ctx->stream, (const uint64_t*)bucketed, (const uint32_t*)pl.off, bm, cap, out);
    check_launch("skeys_bucket_sort");
    tr.mark("screen keys: placed");
}

// both at once (callers with nothing else to read back: a 4-byte copy in between)
void sorted_screen_keys(skh_ctx* ctx, const ScreenKeysIn& in, uint64_t n, uint64_t marker_lo, uint64_t marker_hi, uint64_t* out, PendingSort* own) {
    if (n == 0 || in.ng == 0) return;
    ScreenKeysPlan pl;
    uint32_t* d_max = ctx->arena.get<uint32_t>(4);
    screen_keys_count(ctx, in, n, marker_lo, marker_hi, own, d_max, pl);
    uint32_t h_max = 0;
    d2h(&h_max, d_max, 4, ctx->stream);
    screen_keys_place(ctx, in, n, pl, h_max, out, own);
}

}  // namespace skh
