// sketch_build.hip -- turns raw seeding output into the device-resident Sketch tables.
//
// Replaces the per-sketch HashMap<u32,u64> + multi_position_storage of types.rs:207-320 and the marker HashSet
// (types.rs:272) with, per genome:
//   position order : p_seed/p_g (+ p_rep = 1 bit per position: its seed occurs more than index_chain_band times in this genome) -- enumeration side
//                    p_g = padded genome coordinate << 1 | canonical (common.h CTG_PAD): 4 bytes instead of (pos, contig|strand)
//   seed order     : s_g = the same records sorted by (mix32(seed), contig, pos)   (mix32 is a bijection: equal hash <=> equal seed)
//   seed index     : one 64-bit entry per distinct seed, hash << 32 | first record in s_g << 8 | multiplicity, stored in
//                    tab = the genome's open-addressing table (2 home slots per distinct seed + slack)        -- probe side
//                    Built by one sort + a prefix-max placement of the hash-ordered entries (place_tables_kernel): no atomics, and a
//                    probe of an absent seed usually ends at the LDS bitmap or at its home slot after one 8-byte read.
//   markers        : sorted unique u64
#include <algorithm>

#include "internal.h"

namespace skh {

__device__ __forceinline__ uint32_t seg_of(const uint64_t* off, uint32_t n_seg, uint64_t i) {  // largest g with off[g] <= i
    uint32_t lo = 0, hi = n_seg;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// Seed order = (genome, mix32(seed), contig, pos).  The device-wide radix sort works on 32-bit keys holding the genome and as
// many leading hash bits as fit beside it (stable, so equal keys stay in position order): four passes over 8-byte records
// instead of six over 12-byte ones.  The full 64-bit keys (genome << 32 | hash) are rebuilt afterwards and the rare runs that
// still mix several hashes under one 32-bit key are put in order in place (fixup_runs_kernel).
// When the position index needs few bits (idx_bits), the hash bits that do not fit the 32-bit key ride in the value's upper
// bits (carry = 1), so the full hash comes back after the sort without a gather: value = low hash bits << idx_bits | index.
__global__ __launch_bounds__(256) void make_seed_keys_kernel(const uint32_t* p_seed, const uint64_t* pos_off, uint32_t ng, uint64_t n, uint32_t hash_bits,
                                                             uint32_t idx_bits, uint32_t carry, uint32_t* keys32, uint32_t* vals) {
    __shared__ uint32_t g0;
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) g0 = seg_of(pos_off, ng, i0);                              // one search per workgroup; its records span few genomes
    __syncthreads();
    const uint64_t i = i0 + threadIdx.x;
    if (i >= n) return;
    uint32_t g = g0;
    while (i >= pos_off[g + 1]) g++;
    const uint32_t h = mix32(p_seed[i]), idx = (uint32_t)(i - pos_off[g]);
    keys32[i] = hash_bits >= 32 ? h : ((g << hash_bits) | (h >> (32u - hash_bits)));
    vals[i] = carry ? (((h & ((1u << (32u - hash_bits)) - 1u)) << idx_bits) | idx) : idx;
}
__global__ __launch_bounds__(256) void full_keys_kernel(const uint32_t* p_seed, const uint64_t* pos_off, uint32_t ng, uint64_t n, uint32_t hash_bits,
                                                        uint32_t idx_bits, uint32_t carry, const uint32_t* keys32, uint32_t* vals, uint64_t* keys) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k32 = keys32[i], v = vals[i];
    const uint32_t g = hash_bits >= 32 ? 0u : k32 >> hash_bits;
    uint32_t h;
    if (hash_bits >= 32) h = k32;
    else if (carry) { h = ((k32 & ((1u << hash_bits) - 1u)) << (32u - hash_bits)) | (v >> idx_bits); vals[i] = v & ((1u << idx_bits) - 1u); }
    else h = mix32(p_seed[pos_off[g] + v]);
    keys[i] = ((uint64_t)g << 32) | h;
}
// One thread per run of equal 32-bit keys: orders the run by (full key, position index).  Runs are almost always one seed
// (already in order); insertion sort costs one pass then.  Long runs that do mix hashes get an in-place heapsort.
__device__ __forceinline__ bool kv_less(uint64_t ka, uint32_t va, uint64_t kb, uint32_t vb) { return ka < kb || (ka == kb && va < vb); }
__global__ __launch_bounds__(256) void fixup_runs_kernel(const uint32_t* keys32, uint64_t n, uint64_t* keys, uint32_t* vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (i > 0 && keys32[i] == keys32[i - 1])) return;
    const uint32_t k32 = keys32[i];
    uint64_t e = i + 1;
    while (e < n && keys32[e] == k32) e++;
    const uint64_t len = e - i;
    if (len == 1) return;
    uint64_t* K = keys + i; uint32_t* V = vals + i;
    bool sorted = true;
    for (uint64_t x = 1; x < len && sorted; x++) sorted = !kv_less(K[x], V[x], K[x - 1], V[x - 1]);
    if (sorted) return;
    if (len <= 64) {
        for (uint64_t x = 1; x < len; x++) {
            const uint64_t kx = K[x]; const uint32_t vx = V[x]; uint64_t y = x;
            while (y > 0 && kv_less(kx, vx, K[y - 1], V[y - 1])) { K[y] = K[y - 1]; V[y] = V[y - 1]; y--; }
            K[y] = kx; V[y] = vx;
        }
        return;
    }
    auto sift = [&](uint64_t root, uint64_t end) {                                   // max-heap on (key, val)
        for (;;) {
            uint64_t c = 2 * root + 1;
            if (c >= end) break;
            if (c + 1 < end && kv_less(K[c], V[c], K[c + 1], V[c + 1])) c++;
            if (!kv_less(K[root], V[root], K[c], V[c])) break;
            const uint64_t tk = K[root]; K[root] = K[c]; K[c] = tk; const uint32_t tv = V[root]; V[root] = V[c]; V[c] = tv;
            root = c;
        }
    };
    for (uint64_t st = len / 2; st-- > 0;) sift(st, len);
    for (uint64_t end = len - 1; end > 0; end--) {
        const uint64_t tk = K[0]; K[0] = K[end]; K[end] = tk; const uint32_t tv = V[0]; V[0] = V[end]; V[end] = tv;
        sift(0, end);
    }
}

// (pos, contig << 1 | canonical) -> padded coordinate << 1 | canonical
__global__ __launch_bounds__(256) void pack_positions_kernel(const uint32_t* pos, const uint32_t* cc, const uint64_t* pos_off, const uint64_t* ctg_off, uint32_t ng,
                                                             uint64_t n, const uint32_t* goff, uint32_t* p_g) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = seg_of(pos_off, ng, i), c = cc[i];
    p_g[i] = ((goff[ctg_off[g] + g + (c >> 1)] + pos[i]) << 1) | (c & 1u);
}
__global__ __launch_bounds__(256) void unpack_positions_kernel(const uint32_t* p_g, const uint64_t* pos_off, const uint64_t* ctg_off, uint32_t ng, uint64_t p0,
                                                               uint64_t n, const uint32_t* goff, uint32_t* pos, uint32_t* cc) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = seg_of(pos_off, ng, p0 + i), v = p_g[p0 + i];
    const uint32_t* go = goff + ctg_off[g] + g;
    const uint32_t c = ctg_of(go, (uint32_t)(ctg_off[g + 1] - ctg_off[g]), v >> 1);
    if (pos) pos[i] = (v >> 1) - go[c];
    if (cc) cc[i] = (c << 1) | (v & 1u);
}

__global__ __launch_bounds__(256) void head_flags_kernel(const uint64_t* keys, uint64_t n, uint32_t* head) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t* src, const uint64_t* idx, uint32_t n, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// ---- from the sorted records to the sketch tables, in tiles of 1024 records (256 threads x 4 consecutive records)
constexpr uint32_t BT = 1024;
// distinct (genome, hash) keys that START inside each tile
__global__ __launch_bounds__(256) void tile_heads_kernel(const uint64_t* keys, uint64_t n, uint32_t* tile_cnt) {
    __shared__ uint32_t lds[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * BT + 4u * threadIdx.x;
    uint32_t c = 0;
    if (i0 < n) {
        uint64_t prev = i0 ? keys[i0 - 1] : ~keys[0];
        for (uint32_t j = 0; j < 4 && i0 + j < n; j++) { const uint64_t k = keys[i0 + j]; c += k != prev ? 1u : 0u; prev = k; }
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
// number of distinct keys before each genome's first record (one wave per genome boundary)
__global__ __launch_bounds__(256) void genome_dist_off_kernel(const uint64_t* keys, const uint32_t* tile_off, const uint64_t* pos_off, uint32_t ng, uint32_t* out) {
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g > ng) return;
    const uint64_t x = pos_off[g], t0 = x / BT * BT;
    uint32_t c = 0;
    for (uint64_t i = t0 + lane_id(); i < x; i += 64) c += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    c = wave_sum(c);
    if (lane_id() == 0) out[g] = c + tile_off[x / BT];
}
// One pass over the sorted records emits the index entry of every distinct seed (hash | first record | multiplicity; compact, in hash
// order -- place_tables_kernel spreads them over the genome's table), the hash-order position array, and the per-position
// "repetitive seed" bit.  (Was: head flags + a device-wide scan over all records + three more passes.)
__global__ __launch_bounds__(256) void emit_tables_kernel(const uint64_t* keys, const uint32_t* vals, uint64_t n, const uint32_t* tile_off, const uint64_t* pos_off,
                                                          const uint32_t* p_g, uint64_t* ent, uint32_t* s_g, uint32_t* p_rep, uint32_t band) {
    constexpr int R = BT / 256;
    __shared__ uint32_t lds_scan[R * 4];
    __shared__ uint32_t run_cnt[BT + 1];                     // multiplicity of the run that starts at local distinct index x (slot BT: the run cut by the tile start)
    const uint64_t t0 = (uint64_t)blockIdx.x * BT;
    const uint32_t wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    // record (r, thread) = t0 + 256 r + thread: every load below is coalesced and the four rounds' loads are independent
    uint64_t k[R], kp[R]; uint32_t head[R], incl[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint64_t i = t0 + 256u * (uint32_t)r + threadIdx.x;
        k[r] = i < n ? keys[i] : 0; kp[r] = (i > 0 && i < n) ? keys[i - 1] : 0;
        head[r] = (i < n && (i == 0 || k[r] != kp[r])) ? 1u : 0u;
    }
#pragma unroll
    for (int r = 0; r < R; r++) { incl[r] = wave_incl_scan(head[r]); if (l == 63) lds_scan[r * 4 + wv] = incl[r]; }
    // the run that reaches into this tile from the previous one: its multiplicity, found by one thread (runs are short; the
    // count saturates at 65535)
    if (threadIdx.x == 0 && t0 < n && !head[0]) {
        uint64_t b = t0; uint32_t c = 0;
        while (b > 0 && keys[b - 1] == k[0] && c < 65535u) { b--; c++; }
        uint64_t e = t0; while (e < n && keys[e] == k[0] && c < 65535u) { e++; c++; }
        run_cnt[BT] = c;
    }
    __syncthreads();
    const uint32_t d0 = tile_off[blockIdx.x];
    uint32_t lidx[R], base = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) { const uint32_t x = lds_scan[r * 4 + q]; if (q < wv) before += x; tot += x; }
        lidx[r] = base + before + incl[r];                    // heads up to and including this record (this record's run = lidx - 1; 0 = the cut run)
        base += tot;
        if (head[r]) {
            const uint64_t i = t0 + 256u * (uint32_t)r + threadIdx.x;
            uint32_t c = 1; while (i + c < n && keys[i + c] == k[r] && c < 65535u) c++;
            run_cnt[lidx[r] - 1] = c;
            const uint32_t g = (uint32_t)(k[r] >> 32), hash = (uint32_t)k[r];
            const uint64_t d = (uint64_t)d0 + lidx[r] - 1;
            // one 8-byte entry answers a probe completely: hash | first record in the hash-order array | multiplicity
            ent[d] = ((uint64_t)hash << 32) | ((uint64_t)((uint32_t)(i - pos_off[g]) & 0xFFFFFFu) << 8) | (c > 255u ? 255u : c);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint64_t i = t0 + 256u * (uint32_t)r + threadIdx.x;
        if (i < n) {
            const uint32_t g = (uint32_t)(k[r] >> 32);
            const uint64_t src = pos_off[g] + vals[i];
            const uint32_t c = run_cnt[lidx[r] ? lidx[r] - 1 : BT];
            s_g[i] = p_g[src];
            if (c > band) atomicOr(&p_rep[src >> 5], 1u << (src & 31u));             // rare: the join skips these positions (chain.rs:674-676)
        }
    }
}

// Seed table of a genome: open addressing with the entries themselves in the slots, so that a probe costs ONE memory request
// (one 64-byte line) where a directory + entry array cost two dependent ones.  The entries arrive sorted by hash, and a monotone
// bucket function (common.h seed_bucket) makes their home slots ascend with them: linear-probing placement is then simply
//     slot_i = max(home_i, slot_{i-1} + 1)        <=>        slot_i - i = running maximum of (home_i - i),
// a prefix-max scan -- no atomics, no retries, and every cluster stays sorted by hash, so a probe walks from its home slot while the
// slot's hash is smaller than its own (an empty slot is all ones and ends every walk).  One workgroup per genome writes the whole table
// densely (entry or empty, ascending) and collects the bucket-occupancy bitmap (bit b = some entry's home is b: staged in LDS by the join to
// answer most absent seeds without a memory request) in LDS.  The table has n_buckets + slack slots; an entry pushed beyond them -- tens
// of thousands of seeds hashing into the last buckets -- raises `err`.
constexpr uint32_t PLACE_THREADS = 1024;
__global__ __launch_bounds__(1024) void place_tables_kernel(const uint64_t* __restrict__ ent, const uint64_t* __restrict__ dist_off, const uint64_t* __restrict__ tab_off,
                                                            const uint32_t* __restrict__ n_buckets, const uint64_t* __restrict__ bmap_off, uint32_t lds_words,
                                                            uint64_t* __restrict__ tab, uint32_t* __restrict__ bmap, uint32_t* __restrict__ err) {
    __shared__ int32_t wmax[PLACE_THREADS / 64];
    SKH_DYN_SMEM(smem);
    uint32_t* lbm = (uint32_t*)smem;
    const uint32_t g = blockIdx.x, tid = threadIdx.x, l = tid & 63u, w = tid >> 6;
    const uint64_t e0 = dist_off[g]; const uint32_t nd = (uint32_t)(dist_off[g + 1] - e0), nbk = n_buckets[g];
    uint64_t* T = tab + tab_off[g]; const uint32_t L = (uint32_t)(tab_off[g + 1] - tab_off[g]);
    uint32_t* gbm = bmap + bmap_off[g]; const uint32_t bm_words = (uint32_t)(bmap_off[g + 1] - bmap_off[g]);
    const bool in_lds = bm_words <= lds_words;
    if (in_lds) for (uint32_t x = tid; x < bm_words; x += PLACE_THREADS) lbm[x] = 0;      // (the bitmap in memory was zeroed by the host before the launch)
    __syncthreads();
    constexpr int32_t NEG = -(1 << 30);
    constexpr uint32_t K = 4;                                                        // consecutive entries per thread and round
    int32_t carry = NEG;                                                             // running maximum of home - index over all earlier entries
    for (uint32_t base = 0; base < nd; base += PLACE_THREADS * K) {
        const uint32_t i0 = base + tid * K;
        uint64_t x[K]; uint32_t home[K]; int32_t m[K];
#pragma unroll
        for (uint32_t j = 0; j < K; j++) x[j] = i0 + j < nd ? ent[e0 + i0 + j] : 0;
        int32_t run = NEG;
#pragma unroll
        for (uint32_t j = 0; j < K; j++) {
            home[j] = seed_bucket((uint32_t)(x[j] >> 32), nbk);
            const int32_t v = i0 + j < nd ? (int32_t)home[j] - (int32_t)(i0 + j) : NEG;
            run = v > run ? v : run; m[j] = run;                                     // maximum over this thread's entries up to j
        }
        int32_t v = run;                                                             // inclusive maximum over the wave's threads up to this one
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(v, d, 64); if (l >= (uint32_t)d) v = t > v ? t : v; }
        if (l == 63) wmax[w] = v;
        __syncthreads();
        int32_t pre = carry, all = carry;
        for (uint32_t q = 0; q < PLACE_THREADS / 64; q++) { const int32_t t = wmax[q]; if (q < w) pre = t > pre ? t : pre; all = t > all ? t : all; }
        int32_t before = __shfl_up(v, 1, 64); if (l == 0) before = NEG;               // maximum over the wave's earlier threads
        before = before > pre ? before : pre;                                        // ... and everything before the wave
#pragma unroll
        for (uint32_t j = 0; j < K; j++) {
            const uint32_t i = i0 + j;
            if (i < nd) {
                const int32_t u = m[j] > before ? m[j] : before;                         // u_i
                const int32_t up = j ? (m[j - 1] > before ? m[j - 1] : before) : before;   // u_{i-1}
                const uint32_t slot = (uint32_t)(u + (int32_t)i);
                const uint32_t first = i == 0 ? 0u : (uint32_t)(up + (int32_t)i);       // slot_{i-1} + 1
                if (slot + 1u >= L) atomicAdd(err, 1u);                              // the last slot stays empty: walks end inside the table
                else {
                    for (uint32_t sft = first; sft < slot; sft++) T[sft] = TAB_EMPTY;
                    T[slot] = x[j];
                    if (in_lds) atomicOr(&lbm[home[j] >> 5], 1u << (home[j] & 31u)); else atomicOr(&gbm[home[j] >> 5], 1u << (home[j] & 31u));
                }
            }
        }
        carry = all;
        __syncthreads();
    }
    const uint32_t tail = nd ? (uint32_t)(carry + (int32_t)(nd - 1)) + 1u : 0u;      // first slot after the last entry
    for (uint32_t sft = tail + tid; sft < L; sft += PLACE_THREADS) T[sft] = TAB_EMPTY;
    if (in_lds) { for (uint32_t x = tid; x < bm_words; x += PLACE_THREADS) gbm[x] = lbm[x]; }
}

static int bits_for(uint64_t n) { int b = 1; while ((1ull << b) < n && b < 63) b++; return b; }

void unpack_positions(skh_ctx* ctx, const skh_sketch_set* ss, uint64_t p0, uint64_t n, uint32_t* pos, uint32_t* cc) {
    if (!n || (!pos && !cc)) return;
    SKH_LAUNCH(unpack_positions_kernel, (unsigned)((n + 255) / 256), 256, 0, ctx->stream, (const uint32_t*)ss->p_g.p, (const uint64_t*)ss->d_pos_off.p,
               (const uint64_t*)ss->d_ctg_off.p, ss->n_genomes, p0, n, (const uint32_t*)ss->d_goff.p, pos, cc);
    check_launch("unpack_positions");
}

void build_sketch_tables(skh_ctx* ctx, skh_sketch_set* ss, const uint32_t* pos, const uint32_t* cc) {
    const uint32_t ng = ss->n_genomes;
    const uint64_t P = ss->pos_off[ng];
    StageTrace tr(ctx);
    ss->d_pos_off.alloc(ng + 1); h2d(ss->d_pos_off.p, ss->pos_off.data(), (ng + 1) * 8, ctx->stream);
    ss->d_goff.alloc(ss->goff.size() ? ss->goff.size() : 1); h2d(ss->d_goff.p, ss->goff.data(), ss->goff.size() * 4, ctx->stream);
    ss->d_ctg_off.alloc(ng + 1); h2d(ss->d_ctg_off.p, ss->ctg_off.data(), (ng + 1) * 8, ctx->stream);
    ss->s_g.alloc(P); ss->p_rep.alloc(P / 32 + 1); dzero(ss->p_rep.p, (P / 32 + 1) * 4, ctx->stream);
    if (pos || cc || ss->p_g.n != P) ss->p_g.alloc(P);
    if (P > 0 && pos && cc) {
        SKH_LAUNCH(pack_positions_kernel, (unsigned)((P + 255) / 256), 256, 0, ctx->stream, pos, cc, (const uint64_t*)ss->d_pos_off.p,
                   (const uint64_t*)ss->d_ctg_off.p, ng, P, (const uint32_t*)ss->d_goff.p, ss->p_g.p);
        check_launch("pack_positions");
    }
    ss->dist_off.assign(ng + 1, 0);
    for (uint32_t g = 0; g < ng; g++)
        if (ss->pos_off[g + 1] - ss->pos_off[g] >= (1ull << 24)) throw Error("a genome with >= 2^24 seed positions does not fit the 24-bit table slot field");
    uint64_t D = 0;
    const uint64_t* sorted_keys = nullptr; const uint32_t* sorted_vals = nullptr; const uint32_t* sorted_tile_off = nullptr; uint32_t n_sorted_tiles = 0;
    if (P > 0) {
        if (P >= 0xFFFFFFF0ull) throw Error("sketch set too large for one build (>= 2^32 seed positions); split the batch");
        uint64_t* keys = ctx->arena.get<uint64_t>(P); uint32_t* vals = ctx->arena.get<uint32_t>(P); uint32_t* keys32 = ctx->arena.get<uint32_t>(P);
        const unsigned nb = (unsigned)((P + 255) / 256);
        const uint32_t gbits = ng > 1 ? (uint32_t)bits_for(ng) : 0u;
        const uint32_t hash_bits = std::max<uint32_t>(ctx->tune.build_hash_bits ? std::min<uint32_t>(ctx->tune.build_hash_bits, 32u - gbits) : 32u - gbits, 1u);
        if (gbits >= 32) throw Error("too many genomes in one sketch set");
        uint64_t max_pos = 1; for (uint32_t g = 0; g < ng; g++) max_pos = std::max<uint64_t>(max_pos, ss->pos_off[g + 1] - ss->pos_off[g]);
        const uint32_t idx_bits = (uint32_t)bits_for(max_pos);
        const uint32_t carry = (hash_bits < 32 && idx_bits + (32u - hash_bits) <= 32u && !ctx->tune.build_hash_bits) ? 1u : 0u;
        SKH_LAUNCH(make_seed_keys_kernel, nb, 256, 0, ctx->stream, (const uint32_t*)ss->p_seed.p, (const uint64_t*)ss->d_pos_off.p, ng, P, hash_bits, idx_bits, carry, keys32, vals);
        check_launch("make_seed_keys");
        tr.mark("build: allocs + keys");
        sort_pairs_u32_u32(ctx, keys32, vals, P, (int)(hash_bits >= 32 ? 32 : hash_bits + gbits));
        SKH_LAUNCH(full_keys_kernel, nb, 256, 0, ctx->stream, (const uint32_t*)ss->p_seed.p, (const uint64_t*)ss->d_pos_off.p, ng, P, hash_bits, idx_bits, carry, (const uint32_t*)keys32, vals, keys);
        check_launch("full_keys");
        // always: besides separating hashes that share a 32-bit key it puts equal seeds into position order, which must not
        // depend on the device sort being stable (rocPRIM's path for mid-sized inputs is not)
        SKH_LAUNCH(fixup_runs_kernel, nb, 256, 0, ctx->stream, (const uint32_t*)keys32, P, keys, vals);
        check_launch("fixup_runs");
        tr.mark("build: sort");
        const uint32_t n_bt = (uint32_t)((P + BT - 1) / BT);
        uint32_t* tile_cnt = ctx->arena.get<uint32_t>(n_bt); uint32_t* tile_off = ctx->arena.get<uint32_t>(n_bt + 1);
        SKH_LAUNCH(tile_heads_kernel, n_bt, 256, 0, ctx->stream, (const uint64_t*)keys, P, tile_cnt);
        check_launch("tile_heads");
        exclusive_scan_u32(ctx, tile_cnt, n_bt, tile_off);
        uint32_t* d_do = ctx->arena.get<uint32_t>(ng + 1);
        SKH_LAUNCH(genome_dist_off_kernel, (ng + 1 + 3) / 4, 256, 0, ctx->stream, (const uint64_t*)keys, (const uint32_t*)tile_off, (const uint64_t*)ss->d_pos_off.p, ng, d_do);
        check_launch("genome_dist_off");
        std::vector<uint32_t> h_do(ng + 1);
        d2h(h_do.data(), d_do, (ng + 1) * 4, ctx->stream);
        for (uint32_t g = 0; g <= ng; g++) ss->dist_off[g] = h_do[g];
        D = ss->dist_off[ng];
        tr.mark("build: heads + scan + readback");
        sorted_keys = keys; sorted_vals = vals; sorted_tile_off = tile_off; n_sorted_tiles = n_bt;
    }
    uint64_t* ent = ctx->arena.get<uint64_t>(D + 1);                                 // compact entries, hash order: input of the placement only
    // seed tables (north-star requirement: per-sketch seed -> position tables built on device): 2 buckets per distinct seed + slack
    ss->tab_off.assign(ng + 1, 0); ss->n_buckets.assign(ng, 0); ss->bmap_off.assign(ng + 1, 0);
    uint64_t max_bm_words = 0;
    for (uint32_t g = 0; g < ng; g++) {
        const uint64_t dg = ss->dist_off[g + 1] - ss->dist_off[g];
        ss->n_buckets[g] = (uint32_t)std::max<uint64_t>(16, 2 * dg);                  // dg < 2^24
        ss->tab_off[g + 1] = ss->tab_off[g] + ss->n_buckets[g] + std::max<uint64_t>(64, dg / 8);
        const uint64_t bw = (((uint64_t)ss->n_buckets[g] + 31) / 32 + 3) / 4 * 4;       // whole 16-byte groups
        ss->bmap_off[g + 1] = ss->bmap_off[g] + bw; max_bm_words = std::max(max_bm_words, bw);
    }
    ss->tab.alloc(ss->tab_off[ng] ? ss->tab_off[ng] : 1);
    ss->d_dist_off.alloc(ng + 1); h2d(ss->d_dist_off.p, ss->dist_off.data(), (ng + 1) * 8, ctx->stream);
    ss->d_n_buckets.alloc(ng ? ng : 1); h2d(ss->d_n_buckets.p, ss->n_buckets.data(), ng * 4, ctx->stream);
    if (P > 0) {
        SKH_LAUNCH(emit_tables_kernel, n_sorted_tiles, 256, 0, ctx->stream, sorted_keys, sorted_vals, P, sorted_tile_off, (const uint64_t*)ss->d_pos_off.p,
                   (const uint32_t*)ss->p_g.p, ent, ss->s_g.p, ss->p_rep.p, BP_CHAIN_BAND / ss->params.c);
        check_launch("emit_tables");
    }
    tr.mark("build: entries + gather");
    const uint64_t BW = ss->bmap_off[ng];
    ss->bmap.alloc(BW ? BW : 1);
    uint32_t h_err = 0;
    if (ng) {
        uint64_t* d_to = ctx->arena.get<uint64_t>(ng + 1); h2d(d_to, ss->tab_off.data(), (ng + 1) * 8, ctx->stream);
        uint64_t* d_bo = ctx->arena.get<uint64_t>(ng + 1); h2d(d_bo, ss->bmap_off.data(), (ng + 1) * 8, ctx->stream);
        uint32_t* d_err = ctx->arena.get<uint32_t>(1); dzero(d_err, 4, ctx->stream);
        const uint32_t lds_words = (uint32_t)std::min<uint64_t>(max_bm_words, ctx->tune.place_lds_words);   // LDS for the bitmap; longer ones are set in memory
        dzero(ss->bmap.p, BW * 4, ctx->stream);
        SKH_LAUNCH(place_tables_kernel, ng, PLACE_THREADS, (size_t)lds_words * 4, ctx->stream, (const uint64_t*)ent, (const uint64_t*)ss->d_dist_off.p, (const uint64_t*)d_to,
                   (const uint32_t*)ss->d_n_buckets.p, (const uint64_t*)d_bo, lds_words, ss->tab.p, ss->bmap.p, d_err);
        check_launch("place_tables");
        d2h(&h_err, d_err, 4, ctx->stream);                                            // synchronises
    }
    tr.mark("build: tables");
    dsync(ctx->stream);
    if (h_err) throw Error("seed table overflow: a genome's seeds crowd the end of the hash range");
}

// ---- markers: sort by (genome, marker), drop duplicates (marker_seeds is a set: seeding.rs:318, types.rs:272)
__global__ __launch_bounds__(256) void marker_keys_kernel(uint64_t* raw, const uint64_t* raw_off, uint32_t ng, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    raw[i] |= (uint64_t)seg_of(raw_off, ng, i) << 42;
}
__global__ __launch_bounds__(256) void marker_compact_kernel(const uint64_t* keys, const uint32_t* head, const uint32_t* excl, uint64_t n, uint64_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    out[excl[i]] = keys[i] & ((1ull << 42) - 1);
}

void build_markers(skh_ctx* ctx, skh_sketch_set* ss, DBuf<uint64_t>& raw, const std::vector<uint64_t>& raw_off) {
    const uint32_t ng = ss->n_genomes;
    const uint64_t M = raw_off[ng];
    StageTrace tr(ctx);
    ss->mk_off.assign(ng + 1, 0);
    if (ng >= (1u << 22)) throw Error("more than 4M genomes in one sketch set");
    if (M >= 0xFFFFFFF0ull) throw Error("too many markers for one build; split the batch");
    if (M > 0) {
        uint64_t* d_ro = ctx->arena.get<uint64_t>(ng + 1); h2d(d_ro, raw_off.data(), (ng + 1) * 8, ctx->stream);
        const unsigned nb = (unsigned)((M + 255) / 256);
        SKH_LAUNCH(marker_keys_kernel, nb, 256, 0, ctx->stream, raw.p, (const uint64_t*)d_ro, ng, M);
        check_launch("marker_keys");
        sort_keys_u64(ctx, raw.p, M, 42 + bits_for(ng));
        uint32_t* head = ctx->arena.get<uint32_t>(M); uint32_t* excl = ctx->arena.get<uint32_t>(M + 1);
        SKH_LAUNCH(head_flags_kernel, nb, 256, 0, ctx->stream, (const uint64_t*)raw.p, M, head);
        check_launch("head_flags");
        exclusive_scan_u32(ctx, head, M, excl);
        uint32_t* d_mo = ctx->arena.get<uint32_t>(ng + 1);
        SKH_LAUNCH(gather_u32_kernel, (ng + 1 + 255) / 256, 256, 0, ctx->stream, (const uint32_t*)excl, (const uint64_t*)d_ro, ng + 1, d_mo);
        check_launch("gather_u32");
        std::vector<uint32_t> h_mo(ng + 1);
        d2h(h_mo.data(), d_mo, (ng + 1) * 4, ctx->stream);
        for (uint32_t g = 0; g <= ng; g++) ss->mk_off[g] = h_mo[g];
        ss->markers.alloc(ss->mk_off[ng]);
        SKH_LAUNCH(marker_compact_kernel, nb, 256, 0, ctx->stream, (const uint64_t*)raw.p, (const uint32_t*)head, (const uint32_t*)excl, M, ss->markers.p);
        check_launch("marker_compact");
    } else ss->markers.alloc(0);
    ss->d_mk_off.alloc(ng + 1); h2d(ss->d_mk_off.p, ss->mk_off.data(), (ng + 1) * 8, ctx->stream);
    dsync(ctx->stream);
    tr.mark("build: markers");
}

// host-only: per-genome contig statistics used by switch_qr (chain.rs:625-631) and the regression features
// (chain.rs:519-526): sorted contig lengths at indices n*10/100, n*50/100, n*90/100; and the padded contig starts.
void finalize_metadata(skh_sketch_set* ss) {
    const uint32_t ng = ss->n_genomes;
    ss->goff.assign(ss->ctg_len.size() + ng, 0);
    for (uint32_t g = 0; g < ng; g++) {
        uint64_t at = CTG_PAD; const uint64_t base = ss->ctg_off[g] + g, lim = (1ull << 31) - CTG_PAD;
        for (uint64_t c = ss->ctg_off[g]; c < ss->ctg_off[g + 1]; c++) { ss->goff[base + (c - ss->ctg_off[g])] = (uint32_t)at; at += (uint64_t)ss->ctg_len[c] + CTG_PAD; if (at >= lim) break; }
        if (at >= lim) throw Error("a genome spans >= 2^31 padded bases (total length + 8192 per contig); it does not fit the 32-bit position records");
        ss->goff[base + (ss->ctg_off[g + 1] - ss->ctg_off[g])] = (uint32_t)at;
    }
    ss->mean_ctg.assign(ng, 0.); ss->q10.assign(ng, 0.f); ss->q50.assign(ng, 0.f); ss->q90.assign(ng, 0.f);
    for (uint32_t g = 0; g < ng; g++) {
        std::vector<uint32_t> v(ss->ctg_len.begin() + ss->ctg_off[g], ss->ctg_len.begin() + ss->ctg_off[g + 1]);
        if (v.empty()) continue;
        double s = 0; for (auto x : v) s += (double)x;
        ss->mean_ctg[g] = s / (double)v.size();
        std::sort(v.begin(), v.end());
        size_t n = v.size();
        ss->q10[g] = (float)v[n * 10 / 100]; ss->q50[g] = (float)v[n * 50 / 100]; ss->q90[g] = (float)v[n * 90 / 100];
    }
}

}  // namespace skh
