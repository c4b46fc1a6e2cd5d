// pack_seed.hip -- genome ingest (ASCII -> 2-bit) and FracMinHash k-mer seeding on gfx950.
//
// Replaces seeding.rs:225-323 (fmh_seeds) / avx2_seeding.rs:33-272 (avx2_fmh_seeds) as called per contig
// from file_io.rs:194-226.  Data layout in HBM:
//   packed : u32 words, 16 bases per word, MSB first (base b of a contig -> word (base0+b)/16, shift 30-2*(b%16));
//            every contig starts on a 64-base boundary, so a workgroup's tile starts on a 16 B boundary.
//   nmask  : 1 bit per base (LSB first) marking the bytes the selected reference seeding path treats as
//            "N" ('N' and 'n' for the scalar path, 'N' only for the AVX2 path); read only for contigs that
//            have one (ContigDesc::has_n).  BYTE_TO_SEQ maps every non-ACGTU byte to A (types.rs:40-49), so the
//            N information has to travel separately.
// Kernel shape: one workgroup (256 threads) per tile of 8192 consecutive windows of one contig; the tile's
// 2 KB of packed bases (+20-base halo) are staged through LDS with coalesced dword loads; each thread rolls
// the forward / reverse-complement 21-mers over 32 consecutive windows entirely in registers, keeps a 32-bit
// hit mask, and hits (1/c of windows) are re-derived by bit extraction and written in window order after a
// workgroup prefix sum -- so the output is already sorted by (contig, pos) and needs no sort.
#include "internal.h"

namespace skh {

// ------------------------------------------------------------------------------------------------ pack
__device__ __forceinline__ uint32_t base_code(uint32_t b) {   // types.rs:40-49 BYTE_TO_SEQ
    if (b < 4) return b;
    uint32_t l = b | 0x20u;
    if (l == 'c') return 1;
    if (l == 'g') return 2;
    if (l == 't' || l == 'u') return 3;
    return 0;
}

// A thread packs 32 consecutive bases of one contig (a "unit": contigs start on unit boundaries).  Its 32 source bytes are fetched as nine
// 4-byte-aligned words (two four-word loads and one more; neighbouring lanes read neighbouring 32-byte stretches: coalesced) and shifted into place;
// base codes, the validity of every byte (types.rs:40-49: anything but ACGTU / acgtu / 0..3 is an A) and the N flags come out of byte-parallel
// arithmetic on whole words, four bases at a time (round 2's kernel walked its 32 bytes one by one: 9.6 ms per 4.9 Gbases; a variant with
// 16 bases per thread, whose loads cover one contiguous kilobyte per wave, took 5.9 ms against 3.9 -- twice the threads, twice the contig look-ups).
struct __attribute__((packed, aligned(4))) PackWords4 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {                       // 0x80 in every byte of v that is zero (exact, no borrow between bytes)
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}
__device__ __forceinline__ uint32_t shift_bytes(uint32_t hi, uint32_t lo, uint32_t n_bytes) {   // bytes n .. n + 3 of the eight bytes hi:lo
    return n_bytes ? (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8u * n_bytes)) : lo;
}
// four ASCII bases in a word (first base in the lowest byte) -> their codes in one byte, first base in the two highest bits
__device__ __forceinline__ uint32_t codes_of_word(uint32_t w) {
    const uint32_t x = (w >> 1) & 0x03030303u, letter = x ^ ((x >> 1) & 0x01010101u);   // A 0, C 1, G 2, T / U 3 from bits 1..2 of the letter, either case
    const uint32_t lower = w | 0x20202020u;
    const uint32_t is_letter = zero_bytes(lower ^ 0x61616161u) | zero_bytes(lower ^ 0x63636363u) | zero_bytes(lower ^ 0x67676767u) |
                               zero_bytes(lower ^ 0x74747474u) | zero_bytes(lower ^ 0x75757575u);
    const uint32_t is_small = zero_bytes(w & 0xFCFCFCFCu);                          // bytes 0..3 are their own code (the table's first row)
    const uint32_t c = (letter & ((is_letter >> 7) * 3u)) | (w & ((is_small >> 7) * 3u));
    return (c * 0x40100401u) >> 24;                                                 // byte 0 -> bits 7..6, byte 1 -> 5..4, byte 2 -> 3..2, byte 3 -> 1..0
}
// The same for a word of nothing but A C G T U in either case -- the usual word -- in a third of the instructions: the codes from bits 1..3 of the
// letters, then the letter each code stands for is rebuilt (0x41 + {0, 2, 6, 19}) and compared with the upper-cased byte; `bad` collects the
// differences (U differs from the rebuilt T in bit 0, which is let through for code 3 only).  bad != 0 afterwards: some byte of the unit is
// something else (an N, a digit, a byte 0..3 ...) and the unit is redone with codes_of_word / n_flags_of_word.
__device__ __forceinline__ uint32_t codes_of_acgtu_word(uint32_t w, uint32_t& bad) {
    const uint32_t c = ((w >> 1) ^ (w >> 2)) & 0x03030303u;
    const uint32_t b0 = c & 0x01010101u, b1 = (c >> 1) & 0x01010101u, b01 = b0 & b1;
    const uint32_t letter = 0x41414141u + 2u * b0 + 6u * b1 + 11u * b01;
    bad |= ((w & 0xDFDFDFDFu) ^ letter) & ~b01;
    return (c * 0x40100401u) >> 24;
}
__device__ __forceinline__ uint32_t n_flags_of_word(uint32_t w, int mode) {         // bit x = base x is an N for the selected seeding path
    uint32_t f = zero_bytes(w ^ 0x4E4E4E4Eu);                                       // 'N': seeding.rs:272-275 and avx2_seeding.rs:115-126
    if (mode == SKH_SEED_SCALAR) f |= zero_bytes(w ^ 0x6E6E6E6Eu);                  // 'n': the scalar path only
    return ((f >> 7) * 0x10204080u) >> 28;
}
// A wave packs PACK_ROUNDS x 64 consecutive units, 64 per round.  The contig of its first unit comes from one binary search; from there every lane
// walks on (contigs have at least 16 units: a step now and then) and keeps its contig's record in registers, so a round is the nine source words, the
// arithmetic and two stores.  One round per wave -- a search and three dependent look-ups in front of every 2 KB of source -- ran at the latency of those
// look-ups: 3.9 ms per 4.9 Gbases (1.27 TB/s of ASCII); eight rounds: 2.27 ms, and the kernel then ran at the rate of its ~560 vector instructions per
// unit (the exact byte table in byte-parallel arithmetic: 55 per word, + 9 for the N flags) -- hence codes_of_acgtu_word for the words that are all letters.
#ifndef PACK_ROUNDS_N
#define PACK_ROUNDS_N 8
#endif
constexpr uint32_t PACK_ROUNDS = PACK_ROUNDS_N;
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t* bases, uint64_t readable_bytes, const uint64_t* src_off, const uint64_t* unit_off,
                                                   ContigDesc* contigs, uint32_t n_contigs, uint64_t n_units, int mode,
                                                   uint32_t* packed, uint32_t* nmask) {
    const uint32_t l = threadIdx.x & 63u;
    const uint64_t first = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64u * PACK_ROUNDS);   // the wave's first unit
    if (first >= n_units) return;
    uint32_t lo = 0, hi = n_contigs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (unit_off[mid] <= first) lo = mid; else hi = mid; }
    uint32_t ci = lo, cached = 0xFFFFFFFFu, len = 0;
    uint64_t c_unit = 0, c_next = 0, c_src = 0;                                    // the cached contig: first unit, first unit of its successor, first source byte
    // a round's source words are fetched while the round before is computed
    struct Fetched { PackWords4 qa, qb; uint32_t q8, sh, ci, len; uint64_t b0, src; bool live, whole; };
    auto fetch = [&](uint32_t r) {
        Fetched f; f.qa = PackWords4{0, 0, 0, 0}; f.qb = f.qa; f.q8 = 0; f.sh = 0; f.whole = false;
        const uint64_t u = first + (uint64_t)r * 64u + l;
        f.live = r < PACK_ROUNDS && u < n_units;
        if (!f.live) { f.ci = cached; f.len = 0; f.b0 = 0; f.src = 0; return f; }
        if (cached != ci || u >= c_next) {
            while (ci + 1 < n_contigs && unit_off[ci + 1] <= u) ci++;
            cached = ci; c_unit = unit_off[ci]; c_next = ci + 1 < n_contigs ? unit_off[ci + 1] : ~0ull; c_src = src_off[ci]; len = contigs[ci].len;
        }
        f.ci = ci; f.len = len; f.b0 = (u - c_unit) * 32; f.src = c_src;
        const uint64_t so = c_src + f.b0;                                           // first source byte of the unit
        const uint64_t addr = (uint64_t)(uintptr_t)bases + so; f.sh = (uint32_t)(addr & 3u);
        f.whole = f.b0 + 32 <= len && so - f.sh + 36 <= readable_bytes;             // a whole unit inside the contig, all nine words readable
        if (f.whole) { const uint32_t* q = (const uint32_t*)(uintptr_t)(addr - f.sh); f.qa = *(const PackWords4*)q; f.qb = *(const PackWords4*)(q + 4); f.q8 = q[8]; }
        return f;
    };
    uint32_t n_ci = 0xFFFFFFFFu, seen_n = 0;                                         // N flags seen in contig n_ci
    Fetched cur = fetch(0);
    for (uint32_t r = 0; r < PACK_ROUNDS; r++) {
        const Fetched nxt = fetch(r + 1);
        if (!cur.live) break;
        const uint64_t u = first + (uint64_t)r * 64u + l;
        uint32_t w0 = 0, w1 = 0, m = 0;
        if (cur.whole) {
            const PackWords4 qa = cur.qa, qb = cur.qb; const uint32_t q8 = cur.q8, sh = cur.sh;
            const uint32_t d[8] = {shift_bytes(qa.y, qa.x, sh), shift_bytes(qa.z, qa.y, sh), shift_bytes(qa.w, qa.z, sh), shift_bytes(qb.x, qa.w, sh),
                                   shift_bytes(qb.y, qb.x, sh), shift_bytes(qb.z, qb.y, sh), shift_bytes(qb.w, qb.z, sh), shift_bytes(q8, qb.w, sh)};
            uint32_t bad = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { w0 |= codes_of_acgtu_word(d[j], bad) << (24 - 8 * j); w1 |= codes_of_acgtu_word(d[4 + j], bad) << (24 - 8 * j); }
            if (bad) {                                                              // a byte that is not a base letter: the exact table, and the N flags
                w0 = 0; w1 = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { w0 |= codes_of_word(d[j]) << (24 - 8 * j); w1 |= codes_of_word(d[4 + j]) << (24 - 8 * j); }
#pragma unroll
                for (int j = 0; j < 8; j++) m |= n_flags_of_word(d[j], mode) << (4 * j);
            }
        } else {                                                                    // a contig's last unit(s), the end of the buffer: byte by byte
            const uint8_t* src = bases + cur.src;
            for (uint32_t x = 0; x < 32; x++) {
                const uint64_t p = cur.b0 + x;
                const uint32_t byte = p < cur.len ? src[p] : (uint32_t)'A';
                const uint32_t code = base_code(byte);
                const bool is_n = byte == 78u || (mode == SKH_SEED_SCALAR && byte == 110u);
                if (x < 16) w0 |= code << (30 - 2 * x); else w1 |= code << (30 - 2 * (x - 16));
                m |= (is_n ? 1u : 0u) << x;
            }
        }
        *(uint2*)(packed + 2 * u) = make_uint2(w0, w1); nmask[u] = m;               // contig bases are laid out at 32 * unit_off
        if (m) {
            if (n_ci != cur.ci && seen_n) atomicOr(&contigs[n_ci].has_n, 1u);
            n_ci = cur.ci; seen_n = 1;
        }
        cur = nxt;
    }
    if (seen_n) atomicOr(&contigs[n_ci].has_n, 1u);
}

static inline uint32_t windows_end(uint32_t len, int mode) {  // exclusive bound on the window's last-base index i
    if (len < 2 * K_MARKER) return K_MARKER - 1;                                  // seeding.rs:242 / avx2_seeding.rs:56
    if (mode == SKH_SEED_AVX2) return (K_MARKER - 1) + 4 * ((len - (K_MARKER - 1)) / 4);   // avx2_seeding.rs:48,108
    return len;                                                                    // seeding.rs:271
}

// A genome set is filled in batches of whole genomes (skh_genomes_begin / _append / _finish; skh_genomes_pack is one batch): the packed arrays are
// sized once for the announced capacity, every batch's ASCII goes to one of two device staging buffers (an asynchronous copy when the caller's
// buffer is pinned) and is packed behind it on the context's stream, so the host parses the next files while the previous ones cross PCIe.
void genomes_begin(skh_ctx* ctx, skh_genome_set* gs, uint64_t max_bases, uint32_t max_contigs, uint32_t n_genomes) {
    const uint64_t cap_units = (max_bases + (uint64_t)max_contigs * (CONTIG_ALIGN - 1)) / 32 + 2 * (uint64_t)max_contigs + 2;
    const uint64_t slack_words = 2 * (SEED_TILE / 16) + 64;    // a tile may read one tile + halo past its contig
    gs->cap_units = cap_units; gs->cap_contigs = max_contigs; gs->n_units = 0; gs->open = true;
    gs->packed.alloc(cap_units * 2 + slack_words); gs->nmask.alloc(cap_units + slack_words / 2 + 2);
    gs->d_contigs.alloc(max_contigs ? max_contigs : 1);
    gs->contigs.clear(); gs->genome_contig_off.assign((size_t)n_genomes + 1, 0); gs->n_genomes = n_genomes; gs->n_contigs = 0; gs->total_bases = 0;
}

// contig i of the batch = bases[contig_start[i] .. + contig_len[i]); contig_genome[i] = its genome's number in the set.  A genome's contigs arrive
// together, in their order, in ONE batch; the genomes themselves may arrive in any order (parser threads finish as they finish).
// Returns after queuing; `copied` (may be null) is recorded behind the copy of `bases`: the caller's buffer is free once it has passed.
void genomes_append(skh_ctx* ctx, skh_genome_set* gs, const uint8_t* bases, const uint64_t* contig_start, const uint64_t* contig_len, const uint32_t* contig_genome,
                    uint32_t nc, int on_device, DevEvent* copied) {
    if (!gs->open) throw std::invalid_argument("the genome set is finished: no further batches");
    if ((uint64_t)gs->n_contigs + nc > 0xFFFFFFF0ull) throw std::invalid_argument("too many contigs in one genome set");
    if (gs->n_contigs + nc > gs->cap_contigs) {                                     // max_contigs was an estimate (a parser does not know it beforehand): grow
        const uint32_t cap = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, std::max<uint64_t>((uint64_t)gs->cap_contigs * 2, (uint64_t)gs->n_contigs + nc));
        dsync(ctx->stream);                                                         // queued pack kernels set has_n flags in the old array
        DBuf<ContigDesc> bigger(cap);
        d2d(bigger.p, gs->d_contigs.p, (size_t)gs->n_contigs * sizeof(ContigDesc), ctx->stream);
        dsync(ctx->stream);
        gs->d_contigs = std::move(bigger); gs->cap_contigs = cap;
    }
    const uint32_t c0 = gs->n_contigs, ng = gs->n_genomes;
    std::vector<uint64_t> unit_off(nc + 1, 0), src_off(nc);
    uint64_t span_lo = ~0ull, span_hi = 0;                                        // the stretch of the caller's buffer the batch reads
    // everything that can refuse the batch is checked before the set is touched: a refused append leaves the set as it was
    {
        uint64_t units = 0;
        std::vector<uint8_t> seen;                                                   // genomes that started a run earlier in THIS batch (one flag per genome: a batch may be a whole collection)
        for (uint32_t i = 0; i < nc; i++) {
            const uint32_t g = contig_genome[i];
            if (g >= ng) throw std::invalid_argument("contig_genome must be < n_genomes");
            if (i == 0 || contig_genome[i - 1] != g) {
                if (gs->genome_contig_off[(size_t)g + 1]) throw std::invalid_argument("the contigs of a genome must arrive together, in one batch");
                if (seen.empty()) seen.assign(ng, 0);
                if (seen[g]) throw std::invalid_argument("the contigs of a genome must arrive together, in one batch");
                seen[g] = 1;
            }
            if (contig_len[i] > 0xFFFFFFF0ull) throw std::invalid_argument("contig longer than 2^32 bases");
            units += (contig_len[i] + CONTIG_ALIGN - 1) / CONTIG_ALIGN * CONTIG_ALIGN / 32;
        }
        if (gs->n_units + units > gs->cap_units) throw std::invalid_argument("more bases than skh_genomes_begin announced (every contig takes its length rounded up to 64 bases)");
    }
    for (uint32_t i = 0; i < nc; i++) gs->genome_contig_off[(size_t)contig_genome[i] + 1]++;
    gs->contigs.resize((size_t)c0 + nc);
    uint32_t idx = 0; uint64_t gat = 0;
    for (uint32_t i = 0; i < nc; i++) {
        const uint32_t g = contig_genome[i];
        if (i == 0 || contig_genome[i - 1] != g) idx = 0;
        const uint64_t len = contig_len[i];
        ContigDesc& cd = gs->contigs[c0 + i];
        cd.genome = g; cd.index = idx++;
        cd.len = (uint32_t)len; cd.base = (gs->n_units + unit_off[i]) * 32; cd.has_n = 0;
        if (cd.index == 0) gat = CTG_PAD;
        cd.goff = (uint32_t)gat; cd.goff_hi = (uint32_t)(gat >> 32); gat += len + CTG_PAD;   // (a genome beyond 2^31 makes its sketch set wide: finalize_metadata)
        src_off[i] = contig_start[i];
        if (len) { span_lo = std::min(span_lo, contig_start[i]); span_hi = std::max(span_hi, contig_start[i] + len); }
        const uint64_t padded = (len + CONTIG_ALIGN - 1) / CONTIG_ALIGN * CONTIG_ALIGN;
        unit_off[i + 1] = unit_off[i] + padded / 32;
        gs->total_bases += len;
    }
    const uint64_t n_units = unit_off[nc];
    if (span_lo > span_hi) { span_lo = 0; span_hi = 0; }
    h2d(gs->d_contigs.p + c0, gs->contigs.data() + c0, (size_t)nc * sizeof(ContigDesc), ctx->stream);
    // bytes the kernel may read from the source: rounded up to whole 4-byte words at both ends (an aligned word that holds a valid byte is readable)
    const uint8_t* d_bases = bases; uint64_t readable;
    if (!on_device) {
        DBuf<uint8_t>& st = gs->stage[gs->n_batches & 1];                           // the batch before last has been packed: the stream is in order
        const uint64_t need = span_hi - span_lo + 64;
        if (st.n < need) { dsync(ctx->stream); st.alloc(need + need / 4); }
        h2d_big(st.p, bases + span_lo, span_hi - span_lo, ctx->stream);
        for (uint32_t i = 0; i < nc; i++) src_off[i] -= span_lo;
        d_bases = st.p; readable = need;
    } else {
        const uint64_t mis = (uint64_t)(uintptr_t)bases & 3u;
        readable = (span_hi + mis + 3) / 4 * 4 - mis;
    }
    if (copied) copied->record(ctx->stream);
    uint64_t* d_src = ctx->arena.get<uint64_t>(nc + 1); uint64_t* d_unit = ctx->arena.get<uint64_t>(nc + 1);
    h2d(d_src, src_off.data(), (size_t)nc * 8, ctx->stream); h2d(d_unit, unit_off.data(), ((size_t)nc + 1) * 8, ctx->stream);
    if (n_units) {
        SKH_LAUNCH(pack_kernel, (unsigned)((n_units + 256 * PACK_ROUNDS - 1) / (256 * PACK_ROUNDS)), 256, 0, ctx->stream, d_bases, readable, (const uint64_t*)d_src,
                   (const uint64_t*)d_unit, gs->d_contigs.p + c0, nc, n_units, gs->seeding_mode, gs->packed.p + gs->n_units * 2, gs->nmask.p + gs->n_units);
        check_launch("pack_kernel");
    }
    gs->n_units += n_units; gs->n_contigs += nc; gs->n_batches++;
}

void genomes_finish(skh_ctx* ctx, skh_genome_set* gs) {
    if (!gs->open) return;
    gs->open = false;
    const uint32_t nc = gs->n_contigs; const uint64_t n_units = gs->n_units;
    for (size_t g = 0; g + 1 < gs->genome_contig_off.size(); g++) gs->genome_contig_off[g + 1] += gs->genome_contig_off[g];
    const uint64_t slack_words = 2 * (SEED_TILE / 16) + 64;
    gs->n_words = n_units * 2 + slack_words;
    dzero(gs->packed.p + n_units * 2, slack_words * 4, ctx->stream);
    dzero(gs->nmask.p + n_units, (slack_words / 2 + 2) * 4, ctx->stream);
    d2h(gs->contigs.data(), gs->d_contigs.p, (size_t)nc * sizeof(ContigDesc), ctx->stream);   // picks up has_n (syncs)
    gs->stage[0].release(); gs->stage[1].release();
    // contigs in (genome, contig) order: batches may have brought the genomes in any order (within a genome the order is already right)
    bool sorted = true;
    for (uint32_t i = 1; i < nc && sorted; i++) sorted = gs->contigs[i - 1].genome <= gs->contigs[i].genome;
    if (!sorted) {
        std::stable_sort(gs->contigs.begin(), gs->contigs.end(), [](const ContigDesc& a, const ContigDesc& b) { return a.genome < b.genome; });
        h2d(gs->d_contigs.p, gs->contigs.data(), (size_t)nc * sizeof(ContigDesc), ctx->stream);
    }
    // tile list in (genome, contig, window) order
    gs->tiles.clear();
    for (uint32_t i = 0; i < nc; i++) {
        uint32_t iend = windows_end(gs->contigs[i].len, gs->seeding_mode);
        uint32_t nwin = iend - (K_MARKER - 1);
        for (uint32_t t = 0; t * SEED_TILE < nwin; t++) gs->tiles.push_back(SeedTile{i, t});
    }
    // first tile of every genome (tiles are ordered by genome); genomes without tiles point at their successor's
    const size_t n_tiles = gs->tiles.size(); const uint32_t ng = gs->n_genomes;
    gs->genome_first_tile.assign(ng + 1, (uint32_t)n_tiles);
    for (size_t t = n_tiles; t-- > 0;) gs->genome_first_tile[gs->contigs[gs->tiles[t].contig].genome] = (uint32_t)t;
    for (uint32_t g = ng; g-- > 0;) gs->genome_first_tile[g] = std::min(gs->genome_first_tile[g], gs->genome_first_tile[g + 1]);
    gs->d_tiles.alloc(gs->tiles.size() ? gs->tiles.size() : 1);
    h2d(gs->d_tiles.p, gs->tiles.data(), gs->tiles.size() * sizeof(SeedTile), ctx->stream);
    gs->d_genome_first_tile.alloc(ng + 1);                                           // (seed_offsets_kernel reads the genomes' offsets out by it)
    h2d(gs->d_genome_first_tile.p, gs->genome_first_tile.data(), ((size_t)ng + 1) * 4, ctx->stream);
    dsync(ctx->stream);
}

// ------------------------------------------------------------------------------------------------ seeding
__device__ __forceinline__ uint64_t rev2_64(uint64_t x) {   // reverse the order of the 32 two-bit groups
    uint64_t y = __brevll(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}

__device__ __forceinline__ uint32_t rev2_32(uint32_t x) {   // reverse the order of the 16 two-bit groups
    uint32_t y = __brev(x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

// The hot loop of the seeding kernel is bound by VALU issue.  Measured on gfx950 (tools/exp/valu_rates.hip, profiles/r01_valu_rates.md): every
// integer instruction it can be built from -- v_mad_u64_u32, v_lshrrev_b64, v_lshl_add_u64, v_cmp_gt_u64, v_alignbit_b32 ... -- issues at the same
// rate, so what counts is the NUMBER of instructions per window.  hipcc turns the multiplications of the Thomas Wang mix into pairs of
// v_mad_u64_u32 glued with v_mov (registers pairs must be even-aligned) and the hit masks into cmp + cndmask + or: 45 instructions per window.
// The helpers in dev.h pin the cheaper forms (25 per window).
// Round 4: the hot loop no longer decides "hash < threshold" exactly.  It computes seed_probe(), a 32-bit quantity from which a SUPERSET of the hits
// follows with one 32-bit compare (probe_is_candidate): the last step of the mix, key + (key << 31), is only carried out on the high word, without the
// carry of the low words -- the true high word is that value or one more -- and a window is a candidate when the larger of the two could be below the
// threshold's high word.  One candidate in ~2^31 is not a hit; the dense pass behind the loop hashes every candidate's seed once more anyway (for the
// marker test) and drops those (seed_tiles_kernel).  Against the exact form this saves the 64-bit add, the 64-bit compare and the add-with-carry that
// collected the per-lane hit bits: the candidates of window j are a wave-wide mask in scalar registers, and a lane's bits are put together from the
// masks' few set bits (1/c of the windows) by scalar code.
// (wave_mask_ge, or_in_lanes, funnel_shr, shl_add_u64, seed_hash, seed_probe: dev.h, "the seeding loop's instructions")
// Candidate test on seed_probe's n.  With t = high word without the carry, the true high word is t or t + 1 (mod 2^32), and h < thr needs
// it <= thr_hi: every hit has (t + 1 mod 2^32) <= thr_hi + 1, i.e. ~n <= thr_hi + 1 with n = ~(t + 1), i.e. n >= ~(thr_hi + 1).  (t = 0xFFFFFFFF with a carry
// wraps to a true high word of 0: t + 1 = 0 passes.)
__device__ __forceinline__ uint32_t probe_limit(uint64_t thr) { const uint32_t th = (uint32_t)(thr >> 32); return th == 0xFFFFFFFFu ? 0u : ~(th + 1u); }   // (c = 1: every window)

// Slow path, taken only by contigs that contain an N: bit j set = window j of this thread is suppressed.
// scalar (seeding.rs:272-275,300): an N/n at p (p >= 20) suppresses windows i in [p, p+k).
// avx2 (avx2_seeding.rs:63-81,115-126,181): lanes are substrings of length len4 = (L-20)/4; only an 'N' seen in the
// lane's main loop (p >= lane_start = l*len4+20) suppresses that lane's windows i in [p, p+21).
__device__ uint32_t n_suppress_mask(const uint32_t* nmask, const ContigDesc& cd, uint32_t i0, uint32_t iend, uint32_t k, int mode) {
    uint32_t out = 0;
    const uint32_t len4 = (cd.len - (K_MARKER - 1)) / 4;
    for (uint32_t j = 0; j < SEED_RUN; j++) {
        uint32_t i = i0 + j;
        if (i >= iend) break;
        uint32_t span, lo;
        if (mode == SKH_SEED_AVX2) { span = K_MARKER; uint32_t l = (i - (K_MARKER - 1)) / len4; lo = l * len4 + (K_MARKER - 1); }
        else { span = k; lo = K_MARKER - 1; }
        uint32_t start = i + 1 >= span ? i + 1 - span : 0;
        if (start < lo) start = lo;
        bool sup = false;
        for (uint32_t p = start; p <= i; p++) { uint64_t gb = cd.base + p; if ((nmask[gb >> 5] >> (gb & 31)) & 1u) { sup = true; break; } }
        if (sup) out |= 1u << j;
    }
    return out;
}

// One hit window, re-derived from the owning thread's 128 packed bits (a0:a1:a2:a3, 52 bases MSB first): window j ends at base 20 + j.
struct HitRecord { uint32_t seed; bool canonical; uint64_t kmer; };
__device__ __forceinline__ HitRecord derive_hit(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t j, uint32_t smask) {
    const uint64_t hi = ((uint64_t)a0 << 32) | a1, lo = ((uint64_t)a2 << 32) | a3;
    const uint32_t s = 86u - 2u * j;                                                   // 128 - 2*(j+21)
    uint64_t ff = s >= 64 ? (hi >> (s - 64)) : ((hi << (64 - s)) | (lo >> s));        // forward 21-mer, newest base lowest (seeding.rs:278-280)
    ff &= (1ull << 42) - 1;
    const uint64_t rr = rev2_64(~ff) >> 22;                                            // reverse complement (seeding.rs:281-283)
    const uint32_t fs = (uint32_t)ff & smask, rs = (uint32_t)rr & smask;               // seeding.rs:288-289
    HitRecord hr;
    hr.canonical = fs < rs;                                                            // seeding.rs:290-296
    hr.seed = hr.canonical ? fs : rs;
    hr.kmer = ff < rr ? ff : rr;                                                       // seeding.rs:311-316
    return hr;
}

constexpr uint32_t SEED_DROP_MAX = 16;   // candidates-that-are-not-hits a tile notes per listing round (more: further rounds)
template <bool K15>   // k = 15 (every preset): one instruction less per window
__global__ __launch_bounds__(256) void seed_tiles_kernel(const uint32_t* __restrict__ packed, const uint32_t* __restrict__ nmask,
                                                         const ContigDesc* __restrict__ contigs, const SeedTile* __restrict__ tiles,
                                                         const uint32_t* __restrict__ tile_ids, uint32_t k, uint64_t thr, uint64_t thr_m,
                                                         int mode, uint32_t cap_s, uint32_t cap_m,
                                                         uint32_t* __restrict__ t_seed, uint16_t* __restrict__ t_loc,
                                                         uint64_t* __restrict__ t_marker, uint32_t* __restrict__ cnt_s,
                                                         uint32_t* __restrict__ cnt_m) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_w[SEED_TILE / 16 + 8];
    __shared__ uint32_t lds_scan[16];
    __shared__ uint32_t lds_nm, lds_ndrop, lds_again;
    __shared__ uint16_t lds_drop[SEED_DROP_MAX];
    SKH_DYN_SMEM(dyn_smem);
    uint16_t* lds_hit = (uint16_t*)dyn_smem;                 // cap_s entries
    const uint32_t tid = threadIdx.x;
    const SeedTile tile = tiles[tile_ids ? tile_ids[blockIdx.x] : blockIdx.x];
    const ContigDesc cd = contigs[tile.contig];
    const uint32_t iend = cd.len < 2 * K_MARKER ? (K_MARKER - 1)
                        : (mode == SKH_SEED_AVX2 ? (K_MARKER - 1) + 4 * ((cd.len - (K_MARKER - 1)) / 4) : cd.len);
    // stage: words covering bases [first*8192, first*8192 + 8192 + 20)
    const uint64_t word0 = (cd.base + (uint64_t)tile.first * SEED_TILE) >> 4;
    for (uint32_t w = tid; w < SEED_TILE / 16 + 2; w += SEED_THREADS) lds_w[w] = packed[word0 + w];
    __syncthreads();
    // this thread: bases [32*tid, 32*tid+52) of the tile = 20 warm-up bases + 32 windows
    const uint32_t a0 = lds_w[2 * tid], a1 = lds_w[2 * tid + 1], a2 = lds_w[2 * tid + 2], a3 = lds_w[2 * tid + 3];
    const uint32_t smask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    // The rolled 21-mers of seeding.rs:278-283 are never materialised in the hot loop.  Both seed candidates are bit fields of the packed bases:
    //   f & mask(2k) = the k newest bases of the window, newest lowest  = bits [s, s+2k) of a0:a1:a2:a3, s = 126 - 2x (x = the window's last base);
    //   r & mask(2k) = complement of the k OLDEST bases, oldest lowest = bits [2(x-20), ..) of the complemented, group-reversed string w2:w1:w0.
    // One funnel shift + one AND each, with compile-time shifts (the loop is fully unrolled).
    const uint32_t w0 = rev2_32(~a0), w1 = rev2_32(~a1), w2 = rev2_32(~a2);
    const uint32_t lim = probe_limit(thr);
    unsigned long long cand[SEED_RUN];                                            // candidates of window j over the wave (scalar registers)
#pragma unroll
    for (uint32_t j = 0; j < SEED_RUN; j++) {
        const uint32_t s = 86u - 2u * j;                                           // 126 - 2*(20 + j)
        uint32_t seed;
        if (K15) {
            // the 30-bit fields are taken left-aligned (bits 2..31): the two bits below them can only decide between EQUAL fields, so
            // min + one shift replaces mask, mask, min
            const uint32_t s2 = s - 2u, t2 = 2u * j - 2u;
            const uint32_t fw = s2 >= 64 ? funnel_shr(a0, a1, s2 - 64) : (s2 >= 32 ? funnel_shr(a1, a2, s2 - 32) : funnel_shr(a2, a3, s2));
            const uint32_t rw = j == 0 ? (w0 << 2) : (t2 < 32 ? funnel_shr(w1, w0, t2) : funnel_shr(w2, w1, t2 - 32));
            seed = (fw < rw ? fw : rw) >> 2;                                       // seeding.rs:288-296
        } else {
            const uint32_t fw = s >= 64 ? funnel_shr(a0, a1, s - 64) : (s >= 32 ? funnel_shr(a1, a2, s - 32) : funnel_shr(a2, a3, s));
            const uint32_t rw = 2 * j < 32 ? funnel_shr(w1, w0, 2 * j) : funnel_shr(w2, w1, 2 * j - 32);
            const uint32_t fs = fw & smask, rs = rw & smask;                       // seeding.rs:288-289
            seed = fs < rs ? fs : rs;                                              // seeding.rs:290-296
        }
        cand[j] = wave_mask_ge(seed_probe(seed), lim);                               // seeding.rs:300, as a superset (see seed_probe)
    }
    // a lane's candidate bits, from the set bits of the 32 masks: 1/c of the windows, ~16 per wave
    uint32_t hits = 0;
    {
#pragma unroll
        for (uint32_t j = 0; j < SEED_RUN; j++) {
            if (cand[j]) or_in_lanes(hits, cand[j], 1u << j);                      // (all its lanes at once: the mask IS the set of lanes)
        }
    }
    const uint32_t i0 = (K_MARKER - 1) + tile.first * SEED_TILE + SEED_RUN * tid;   // i of this thread's window 0
    uint32_t nvalid = iend > i0 ? iend - i0 : 0;
    const uint32_t vmask = nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u);
    hits &= vmask;
    if (cd.has_n) hits &= ~n_suppress_mask(nmask, cd, i0, iend, k, mode);
    // Candidates that are not hits (hash >= thr: one in ~2^31) are found by the dense pass below, which hashes every listed seed anyway; it notes them in
    // lds_drop, their owners clear the bits, and the listing is done again (a handful of tiles per 5 billion windows).
    const uint32_t wv = tid >> 6, ln = tid & 63;
    const uint64_t obase = (uint64_t)blockIdx.x * cap_s, mbase = (uint64_t)blockIdx.x * cap_m;
    uint32_t tot = 0;
    for (;;) {
        // workgroup prefix sum of the hit counts
        const uint32_t c = (uint32_t)__popc(hits);
        uint32_t incl = wave_incl_scan(c);
        if (ln == 63) lds_scan[wv] = incl;
        if (tid == 0) { lds_nm = 0; lds_ndrop = 0; lds_again = 0; }
        __syncthreads();
        uint32_t base = 0; tot = 0;
        for (uint32_t q = 0; q < SEED_THREADS / 64; q++) { uint32_t t = lds_scan[q]; if (q < wv) base += t; tot += t; }
        // Hits are 1/c of the windows, scattered over the lanes: deriving their records lane by lane would keep a whole wave busy for as many
        // rounds as its unluckiest lane has hits.  Instead every thread appends (thread << 5 | window) for its hits to a list in LDS (window order,
        // from the prefix sum) and the list is worked off densely, one hit per thread: seed, strand and -- from one more hash of the seed -- whether
        // the candidate is a hit at all (seeding.rs:300, exactly) and whether the window is a marker (seeding.rs:311-319).  Marker slots are handed out by an
        // LDS counter; their order is irrelevant (marker_seeds is a set, built by sorting: sketch_build.hip).
        // The tile scratch holds cap_s seeds / cap_m markers; a tile that needs more (low-complexity sequence) is re-run by the host with full
        // capacity -- the two counts are exact either way.
        if (tot <= cap_s) {
            uint32_t so = base + incl - c, hm = hits;
            while (hm) { const uint32_t j = (uint32_t)__ffs((int)hm) - 1u; hm &= hm - 1u; lds_hit[so++] = (uint16_t)((tid << 5) | j); }
            __syncthreads();
            for (uint32_t x = tid; x < tot; x += SEED_THREADS) {
                const uint32_t code = lds_hit[x], src = code >> 5;
                const HitRecord hr = derive_hit(lds_w[2 * src], lds_w[2 * src + 1], lds_w[2 * src + 2], lds_w[2 * src + 3], code & 31u, smask);
                const uint64_t h = seed_hash(hr.seed);
                if (h >= thr) { SKH_SEED_DROP_NOTE(); const uint32_t d = atomicAdd(&lds_ndrop, 1u); if (d < SEED_DROP_MAX) lds_drop[d] = (uint16_t)code; continue; }
                t_seed[obase + x] = hr.seed;
                t_loc[obase + x] = (uint16_t)(code | (hr.canonical ? 0x8000u : 0u));        // code = SEED_RUN * thread + window
                if (h < thr_m) { const uint32_t mo = atomicAdd(&lds_nm, 1u); if (mo < cap_m) t_marker[mbase + mo] = hr.kmer; }
            }
        } else {                                                                               // only the counts matter
            uint32_t hm = hits, nm = 0;
            while (hm) {
                const uint32_t j = (uint32_t)__ffs((int)hm) - 1u; hm &= hm - 1u;
                const uint64_t h = seed_hash(derive_hit(a0, a1, a2, a3, j, smask).seed);
                if (h >= thr) { SKH_SEED_DROP_NOTE(); hits &= ~(1u << j); lds_again = 1; }                              // (dropped by its owner at once; the counts are taken again)
                else if (h < thr_m) nm++;
            }
            if (nm) atomicAdd(&lds_nm, nm);
        }
        __syncthreads();
        const uint32_t nd = lds_ndrop;
        if (!nd && !lds_again) break;
        for (uint32_t d = 0; d < nd && d < SEED_DROP_MAX; d++) { const uint32_t code = lds_drop[d]; if ((code >> 5) == tid) hits &= ~(1u << (code & 31u)); }
        __syncthreads();                                                                       // (lds_ndrop / lds_drop are rewritten by the next round)
    }
    __syncthreads();
    if (tid == 0) { cnt_s[blockIdx.x] = tot; cnt_m[blockIdx.x] = lds_nm; }
}

// What lies between the seeding kernel and its compaction, in TWO launches (it was fourteen: an overflow pass, two scans of four launches each, an upload, two gathers,
// two copies -- 70 us of small kernels):
//   * every workgroup takes 1024 tiles: a tile whose counts exceed the capped scratch gets a slot in the full-capacity overflow scratch; the tiles' seed and marker
//     counts are scanned inside the workgroup (loc_*: a tile's offset within its block of 1024) and the block's totals are published;
//   * the workgroup that finishes LAST (a ticket) scans the block totals (blk_*: exclusive, one more entry = the grand total);
//   * seed_got_kernel reads out what the host wants to know: the offsets at the first tile of every genome that starts inside this launch, the totals and the overflow count.
// A tile's offset is loc[t] + blk[t >> 10]: the compaction kernel adds the two itself.
constexpr uint32_t OFFS_T = 1024, OFFS_PER = 8, OFFS_MAX_BLOCKS = OFFS_T * OFFS_PER;    // (a launch covers at most 8 M tiles = 69 G windows: the last workgroup scans OFFS_PER block totals per thread)
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* total, uint32_t* lds /* 16 words */) {
    const uint32_t incl = wave_incl_scan(v), w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    if (l == 63) lds[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t i = 0; i < 16; i++) { const uint32_t t = lds[i]; if (i < w) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}
__global__ __launch_bounds__(1024) void seed_offsets_kernel(const uint32_t* cnt_s, const uint32_t* cnt_m, uint32_t n_tiles, uint32_t cap_s, uint32_t cap_m,
                                                            uint32_t* ovf_idx, uint32_t* ovf_list, uint32_t* n_ovf /* [0] overflow count, [1] ticket: both zero at launch */,
                                                            uint32_t* loc_s, uint32_t* loc_m, uint32_t* blk_s, uint32_t* blk_m /* gridDim.x + 1 entries each */) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t last;
    const uint32_t t = blockIdx.x * OFFS_T + threadIdx.x;
    const uint32_t cs = t < n_tiles ? cnt_s[t] : 0u, cm = t < n_tiles ? cnt_m[t] : 0u;
    if (t < n_tiles) {
        uint32_t idx = 0xFFFFFFFFu;
        if (cs > cap_s || cm > cap_m) { idx = atomicAdd(&n_ovf[0], 1u); ovf_list[idx] = t; }
        ovf_idx[t] = idx;
    }
    uint32_t tot_s, tot_m;
    const uint32_t os = block_excl_scan_1024(cs, &tot_s, lds), om = block_excl_scan_1024(cm, &tot_m, lds);
    if (t < n_tiles) { loc_s[t] = os; loc_m[t] = om; }
    // The last workgroup reads nothing but the block totals, which thread 0 writes and fences itself.  (Round 4's version also read other workgroups' loc_* here --
    // the genomes' first tiles, for the host -- which needed an agent-scope fence by EVERY thread in front of the ticket: 9,440 waves each writing the XCD's L2
    // back took the kernel from 0.04 to 0.20 ms.  Those reads now sit behind a kernel boundary: seed_got_kernel.)
    if (threadIdx.x == 0) {
        blk_s[blockIdx.x] = tot_s; blk_m[blockIdx.x] = tot_m;                         // (totals for now; the last workgroup turns them into offsets)
        __threadfence();
        last = atomicAdd(&n_ovf[1], 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the last workgroup: exclusive scan of the block totals, OFFS_PER blocks per thread (what the other workgroups wrote is read past this CU's L1)
    const uint32_t nb = gridDim.x;
    uint32_t vs[OFFS_PER], vm[OFFS_PER], ss = 0, sm = 0;
    for (uint32_t i = 0; i < OFFS_PER; i++) {
        const uint32_t b = threadIdx.x * OFFS_PER + i;
        vs[i] = b < nb ? __atomic_load_n(&blk_s[b], __ATOMIC_RELAXED) : 0u; vm[i] = b < nb ? __atomic_load_n(&blk_m[b], __ATOMIC_RELAXED) : 0u;
        ss += vs[i]; sm += vm[i];
    }
    uint32_t all_s, all_m;
    uint32_t bs = block_excl_scan_1024(ss, &all_s, lds), bm = block_excl_scan_1024(sm, &all_m, lds);
    for (uint32_t i = 0; i < OFFS_PER; i++) {
        const uint32_t b = threadIdx.x * OFFS_PER + i;
        if (b < nb) { blk_s[b] = bs; blk_m[b] = bm; }
        bs += vs[i]; bm += vm[i];
    }
    if (threadIdx.x == 0) { blk_s[nb] = all_s; blk_m[nb] = all_m; }
}
// the host's numbers, behind the kernel boundary that orders every workgroup's loc_* / blk_* stores: got[g], got[n_genomes + 1 + g] = the seed / marker offset at genome
// g's first tile if that lies in [tile0, tile0 + n_tiles] (else untouched); the tail: totals and overflow count
__global__ __launch_bounds__(256) void seed_got_kernel(const uint32_t* __restrict__ loc_s, const uint32_t* __restrict__ loc_m, const uint32_t* __restrict__ blk_s,
                                                       const uint32_t* __restrict__ blk_m, uint32_t n_blocks, const uint32_t* __restrict__ n_ovf, uint32_t n_tiles,
                                                       const uint32_t* __restrict__ genome_first_tile, uint32_t n_genomes, uint32_t tile0, uint32_t* __restrict__ got /* 2 (n_genomes + 1) + 3 */) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t all_s = blk_s[n_blocks], all_m = blk_m[n_blocks];
    if (g == 0) { uint32_t* tail = got + 2 * (n_genomes + 1); tail[0] = all_s; tail[1] = all_m; tail[2] = n_ovf[0]; }
    if (g > n_genomes) return;
    const uint32_t f = genome_first_tile[g];
    if (f < tile0 || f > tile0 + n_tiles) return;
    const uint32_t lt = f - tile0;
    got[g] = lt == n_tiles ? all_s : loc_s[lt] + blk_s[lt >> 10];
    got[n_genomes + 1 + g] = lt == n_tiles ? all_m : loc_m[lt] + blk_m[lt >> 10];
}

// A QUARTER of a wave per tile (16 lanes: a tile holds ~65 seeds and ~8 markers) copies the tile's records to their final (contig,pos)-ordered place.  WIDE (a
// set with a genome beyond 31-bit coordinates): the coordinates also go out as 64-bit records (o_g64); the table build replaces the 32-bit records of the
// wide genomes by position indices (sketch_build.hip).  The kernel is a chain of three dependent round trips per tile (tile record + offsets, contig record,
// the tile's scratch): with a whole wave per tile it ran 604,000 short waves in 74 generations (0.31 ms, 88 % of the time waiting); four tiles per wave
// make it a quarter of the waves with the same chain.
template <bool WIDE>
__global__ __launch_bounds__(256) void seed_compact_kernel(const SeedTile* __restrict__ tiles, const ContigDesc* __restrict__ contigs,
                                                           uint32_t n_tiles, uint32_t cap_s, uint32_t cap_m, const uint32_t* __restrict__ ovf_idx,
                                                           const uint32_t* __restrict__ t_seed, const uint16_t* __restrict__ t_loc,
                                                           const uint64_t* __restrict__ t_marker, const uint32_t* __restrict__ o_seed2,
                                                           const uint16_t* __restrict__ o_loc2, const uint64_t* __restrict__ o_marker2,
                                                           const uint32_t* __restrict__ cnt_s, const uint32_t* __restrict__ cnt_m,
                                                           const uint32_t* __restrict__ loc_s, const uint32_t* __restrict__ loc_m,
                                                           const uint32_t* __restrict__ blk_s, const uint32_t* __restrict__ blk_m,
                                                           uint32_t* __restrict__ o_seed, uint32_t* __restrict__ o_g,
                                                           uint64_t* __restrict__ o_g64, uint64_t* __restrict__ o_marker) {
    constexpr uint32_t G = 16;                                                       // lanes per tile
    const uint32_t lt = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (lt >= n_tiles) return;
    const uint32_t ln = threadIdx.x & (G - 1u);
    const SeedTile tile = tiles[lt];
    const uint32_t goff = contigs[tile.contig].goff;
    const uint64_t goff64 = ((uint64_t)contigs[tile.contig].goff_hi << 32) | goff;
    const uint32_t s0 = loc_s[lt] + blk_s[lt >> 10], ns = cnt_s[lt], m0 = loc_m[lt] + blk_m[lt >> 10], nm = cnt_m[lt];   // (seed_offsets_kernel)
    const uint32_t ov = ovf_idx[lt];
    const uint32_t* src_seed = ov == 0xFFFFFFFFu ? t_seed + (uint64_t)lt * cap_s : o_seed2 + (uint64_t)ov * SEED_TILE;
    const uint16_t* src_loc = ov == 0xFFFFFFFFu ? t_loc + (uint64_t)lt * cap_s : o_loc2 + (uint64_t)ov * SEED_TILE;
    const uint64_t* src_mk = ov == 0xFFFFFFFFu ? t_marker + (uint64_t)lt * cap_m : o_marker2 + (uint64_t)ov * SEED_TILE;
    // the loads of up to eight rounds (and the markers') are issued before the first store
    const uint32_t pos0 = (K_MARKER - 1) + tile.first * SEED_TILE;
    const uint64_t mk0 = ln < nm ? src_mk[ln] : 0ull;
    for (uint32_t x0 = 0; x0 < ns; x0 += 8 * G) {
        uint32_t loc[8], sd[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t x = x0 + G * (uint32_t)u + ln; loc[u] = x < ns ? src_loc[x] : 0u; sd[u] = x < ns ? src_seed[x] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t x = x0 + G * (uint32_t)u + ln;
            if (x < ns) {
                o_seed[s0 + x] = sd[u];
                const uint32_t pos = pos0 + (loc[u] & 0x1FFFu);                         // pos = index of the window's last base
                if (WIDE) o_g64[s0 + x] = ((goff64 + pos) << 1) | (loc[u] >> 15);
                o_g[s0 + x] = ((goff + pos) << 1) | (loc[u] >> 15);                     // SeedPosition (types.rs:131-138) in padded coordinates (a wide genome's records are replaced by indices later)
            }
        }
    }
    if (ln < nm) o_marker[m0 + ln] = mk0;
    for (uint32_t x = ln + G; x < nm; x += G) o_marker[m0 + x] = src_mk[x];
}

// async_tail: a set seeded in ONE launch returns with its compaction kernel still queued (no wait, the arena not rewound): the caller queues the table
// build behind it and prepares that build's host tables meanwhile; out.tail_pending says so
void seed_genomes(skh_ctx* ctx, skh_genome_set* gs, const skh_sketch_params& sp, SeedOutput& out, bool async_tail, bool wide, const std::function<void()>* meanwhile) {
    out.tail_pending = false; out.wide = wide;
    const uint64_t thr = ~0ull / (uint64_t)sp.c, thr_m = ~0ull / (uint64_t)sp.marker_c;   // seeding.rs:258-259
    const size_t n_tiles = gs->tiles.size();
    const uint32_t ng = gs->n_genomes;
    StageTrace tr(ctx);
    out.pos_off.assign(ng + 1, 0); out.mk_off.assign(ng + 1, 0);
    // capped tile scratch: 4x the expected hits per tile; tiles that exceed it are re-run with full capacity
    const uint32_t cap_s = ctx->tune.seed_tile_cap ? std::min<uint32_t>(SEED_TILE, ctx->tune.seed_tile_cap) : std::min<uint32_t>(SEED_TILE, std::max<uint32_t>(256, 4 * SEED_TILE / sp.c));
    const uint32_t cap_m = std::min<uint32_t>(SEED_TILE, std::max<uint32_t>(64, 4 * SEED_TILE / sp.marker_c));
    const size_t tile_bytes = (size_t)cap_s * 6 + (size_t)cap_m * 8;
    // tile scratch per launch: up to 16 GB (10,000 genomes of 5 Mbp in ONE launch: two launches meant 1.8 ms of copies putting their outputs behind one another, and no
    // compaction left queued behind the call), but never more than a quarter of what the device has to give right now (free + this library's idle and scratch blocks)
    size_t scratch = (size_t)ctx->tune.seed_scratch_bytes;
    if (!ctx->tune.seed_scratch_fixed) scratch = std::min<size_t>(scratch, std::max<size_t>((size_t)2 << 30, (device_memory_free() + ctx->arena.capacity()) / 4));
    const size_t MAX_TILES = std::min<size_t>(std::max<size_t>(1, scratch / tile_bytes), (size_t)OFFS_MAX_BLOCKS * OFFS_T);
    struct Part { DBuf<uint32_t> seed, g; DBuf<uint64_t> g64, mk; uint64_t ns = 0, nm = 0; };
    std::vector<Part> parts;
    const std::vector<uint32_t>& g_first = gs->genome_first_tile;                     // first tile of every genome (tiles are ordered by genome)
    std::vector<uint64_t> g_ns(ng + 1, 0), g_nm(ng + 1, 0);   // running totals at genome starts
    std::vector<std::pair<DevEvent, DevEvent>> evs;                                   // around every launch of the seeding kernel: bench.py's roofline figure
    tr.mark("seed: host tile tables");
    uint64_t base_s = 0, base_m = 0;
    for (size_t t0 = 0; t0 < n_tiles; t0 += MAX_TILES) {
        const uint32_t nt = (uint32_t)std::min(MAX_TILES, n_tiles - t0);
        uint32_t* t_seed = ctx->arena.get<uint32_t>((size_t)nt * cap_s);
        uint16_t* t_loc = ctx->arena.get<uint16_t>((size_t)nt * cap_s);
        uint64_t* t_marker = ctx->arena.get<uint64_t>((size_t)nt * cap_m);
        uint32_t* cnt_s = ctx->arena.get<uint32_t>(nt); uint32_t* cnt_m = ctx->arena.get<uint32_t>(nt);
        const uint32_t nblk = (nt + OFFS_T - 1) / OFFS_T;
        uint32_t* loc_s = ctx->arena.get<uint32_t>(nt); uint32_t* loc_m = ctx->arena.get<uint32_t>(nt);
        uint32_t* blk_s = ctx->arena.get<uint32_t>(nblk + 1); uint32_t* blk_m = ctx->arena.get<uint32_t>(nblk + 1);
        uint32_t* ovf_idx = ctx->arena.get<uint32_t>(nt); uint32_t* ovf_list = ctx->arena.get<uint32_t>(nt);
        uint32_t* n_ovf = ctx->arena.get<uint32_t>(2);                                  // overflow count, the offsets kernel's ticket
        uint32_t* d_got = ctx->arena.get<uint32_t>(2 * ((size_t)ng + 1) + 3);
        dzero(n_ovf, 8, ctx->stream);
        const SeedTile* d_tiles = gs->d_tiles.p + t0;
        evs.emplace_back();
        evs.back().first.record(ctx->stream);
        if (sp.k == 15) SKH_LAUNCH(seed_tiles_kernel<true>, nt, SEED_THREADS, cap_s * 2, ctx->stream, (const uint32_t*)gs->packed.p, (const uint32_t*)gs->nmask.p,
                   (const ContigDesc*)gs->d_contigs.p, d_tiles, (const uint32_t*)nullptr, sp.k, thr, thr_m, gs->seeding_mode, cap_s, cap_m,
                   t_seed, t_loc, t_marker, cnt_s, cnt_m);
        else SKH_LAUNCH(seed_tiles_kernel<false>, nt, SEED_THREADS, cap_s * 2, ctx->stream, (const uint32_t*)gs->packed.p, (const uint32_t*)gs->nmask.p,
                   (const ContigDesc*)gs->d_contigs.p, d_tiles, (const uint32_t*)nullptr, sp.k, thr, thr_m, gs->seeding_mode, cap_s, cap_m,
                   t_seed, t_loc, t_marker, cnt_s, cnt_m);
        check_launch("seed_tiles_kernel");
        evs.back().second.record(ctx->stream);
        tr.mark("seed: tiles kernel");
        // overflow slots, the tiles' offsets and the numbers the host needs (genome boundaries inside this launch, totals, overflow count) in one launch and one read-back
        SKH_LAUNCH(seed_offsets_kernel, nblk, OFFS_T, 0, ctx->stream, (const uint32_t*)cnt_s, (const uint32_t*)cnt_m, nt, cap_s, cap_m, ovf_idx, ovf_list, n_ovf,
                   loc_s, loc_m, blk_s, blk_m);
        check_launch("seed_offsets");
        SKH_LAUNCH(seed_got_kernel, (ng + 1 + 255) / 256, 256, 0, ctx->stream, (const uint32_t*)loc_s, (const uint32_t*)loc_m, (const uint32_t*)blk_s, (const uint32_t*)blk_m, nblk,
                   (const uint32_t*)n_ovf, nt, (const uint32_t*)gs->d_genome_first_tile.p, ng, (uint32_t)t0, d_got);
        check_launch("seed_got");
        if (meanwhile && t0 == 0) { try { (*meanwhile)(); } catch (...) { device_sync_all(); throw; } }   // (the device is busy for ~3 ms per 5 Gbases: the caller's host work belongs here, not in front of the launch; nothing queued may outlive the arena on an error)
        std::vector<uint32_t> got(2 * ((size_t)ng + 1) + 3);
        d2h(got.data(), d_got, got.size() * 4, ctx->stream);
        const uint32_t h_novf = got[2 * ((size_t)ng + 1) + 2];
        tr.mark("seed: scans + readback");
        Part p; p.ns = got[2 * ((size_t)ng + 1)]; p.nm = got[2 * ((size_t)ng + 1) + 1];
        for (uint32_t g = 0; g <= ng; g++) if (g_first[g] >= t0 && g_first[g] <= t0 + nt) { g_ns[g] = base_s + got[g]; g_nm[g] = base_m + got[(size_t)ng + 1 + g]; }
        uint32_t *o_seed2 = nullptr; uint16_t* o_loc2 = nullptr; uint64_t* o_marker2 = nullptr;
        if (h_novf) {   // second pass over the overflowing tiles with worst-case capacity
            o_seed2 = ctx->arena.get<uint32_t>((size_t)h_novf * SEED_TILE); o_loc2 = ctx->arena.get<uint16_t>((size_t)h_novf * SEED_TILE);
            o_marker2 = ctx->arena.get<uint64_t>((size_t)h_novf * SEED_TILE);
            uint32_t* c2 = ctx->arena.get<uint32_t>(2 * (size_t)h_novf);
            if (sp.k == 15) SKH_LAUNCH(seed_tiles_kernel<true>, h_novf, SEED_THREADS, SEED_TILE * 2, ctx->stream, (const uint32_t*)gs->packed.p, (const uint32_t*)gs->nmask.p,
                       (const ContigDesc*)gs->d_contigs.p, d_tiles, (const uint32_t*)ovf_list, sp.k, thr, thr_m, gs->seeding_mode, SEED_TILE, SEED_TILE,
                       o_seed2, o_loc2, o_marker2, c2, c2 + h_novf);
            else SKH_LAUNCH(seed_tiles_kernel<false>, h_novf, SEED_THREADS, SEED_TILE * 2, ctx->stream, (const uint32_t*)gs->packed.p, (const uint32_t*)gs->nmask.p,
                       (const ContigDesc*)gs->d_contigs.p, d_tiles, (const uint32_t*)ovf_list, sp.k, thr, thr_m, gs->seeding_mode, SEED_TILE, SEED_TILE,
                       o_seed2, o_loc2, o_marker2, c2, c2 + h_novf);
            check_launch("seed_tiles_kernel(overflow)");
        }
        p.seed.alloc(p.ns); p.mk.alloc(p.nm);
        p.g.alloc(p.ns); if (wide) p.g64.alloc(p.ns);
#define SKH_COMPACT(W) SKH_LAUNCH(seed_compact_kernel<W>, (nt + 15) / 16, 256, 0, ctx->stream, d_tiles, (const ContigDesc*)gs->d_contigs.p, nt, cap_s, cap_m, \
                   (const uint32_t*)ovf_idx, (const uint32_t*)t_seed, (const uint16_t*)t_loc, (const uint64_t*)t_marker, (const uint32_t*)o_seed2, \
                   (const uint16_t*)o_loc2, (const uint64_t*)o_marker2, (const uint32_t*)cnt_s, (const uint32_t*)cnt_m, (const uint32_t*)loc_s, (const uint32_t*)loc_m, \
                   (const uint32_t*)blk_s, (const uint32_t*)blk_m, p.seed.p, p.g.p, p.g64.p, p.mk.p)
        if (wide) SKH_COMPACT(true); else SKH_COMPACT(false);
#undef SKH_COMPACT
        check_launch("seed_compact_kernel");
        tr.mark("seed: overflow + alloc + compact");
        base_s += p.ns; base_m += p.nm;
        parts.push_back(std::move(p));
        if (async_tail && parts.size() == 1 && t0 + nt >= n_tiles) { out.tail_pending = true; break; }
        dsync(ctx->stream);
        ctx->arena.reset();
    }
    if (meanwhile && n_tiles == 0) (*meanwhile)();                                    // (nothing was launched: the caller's work is still to be done)
    g_ns[ng] = base_s; g_nm[ng] = base_m;
    for (uint32_t g = 0; g <= ng; g++) { out.pos_off[g] = g_ns[g]; out.mk_off[g] = g_nm[g]; }
    const uint64_t NS = out.pos_off[ng], NM = out.mk_off[ng];
    if (parts.size() == 1) {
        out.seed = std::move(parts[0].seed); out.g = std::move(parts[0].g); out.g64 = std::move(parts[0].g64); out.markers_raw = std::move(parts[0].mk);
    } else {
        out.seed.alloc(NS); out.markers_raw.alloc(NM);
        out.g.alloc(NS); if (wide) out.g64.alloc(NS);
        uint64_t so = 0, mo = 0;
        for (auto& p : parts) {
            d2d(out.seed.p + so, p.seed.p, p.ns * 4, ctx->stream);
            d2d(out.g.p + so, p.g.p, p.ns * 4, ctx->stream); if (wide) d2d(out.g64.p + so, p.g64.p, p.ns * 8, ctx->stream);
            d2d(out.markers_raw.p + mo, p.mk.p, p.nm * 8, ctx->stream);
            so += p.ns; mo += p.nm;
        }
        dsync(ctx->stream);
    }
    float ms = 0; for (auto& e : evs) ms += DevEvent::ms(e.first, e.second);           // (every launch was followed by a synchronising read-back)
    ctx->timings.seed_kernel_ms += ms; ctx->timings.seed_kernel_launches += (uint32_t)evs.size();
}

}  // namespace skh
