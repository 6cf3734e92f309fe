// pack_kernels.hip — ASCII records -> filtered 2-bit stream + record-break bitmap (gfx950).
//
// Replaces, for a whole batch at once, the two per-record copies the reference makes before its k-mer loop:
//   filter_out_n(seqrec.seq())   /root/reference/src/utils.rs:459 -> 33-41  (delete every byte that is not an
//                                upper-case A C G T and JOIN the flanks)
//   KSeq::new(&seq, 2)           utils.rs:464 (kmerutils 2-bit packing, A=0 C=1 G=2 T=3)
// and remembers where each record's surviving bases begin, because k-mers never span records
// (a fresh KSeq per record, utils.rs:457-464) but DO span deleted characters.
//
// One pass over the batch: 1 B/base read, 0.25 B/base written.  Stream compaction needs, for every tile, the number
// of bases that survived before it in the same genome; that prefix comes from a decoupled look-back scan
// (Merrill & Garland) whose per-tile state is ONE naturally aligned 8-byte word
//      bits 63:32  count   surviving bases (AGGREGATE: of this tile; INCLUSIVE: of the genome up to and incl. it)
//      bits 31:2   tail    the last min(count,15) of those bases, last base in bits 3:2
//      bits  1:0   status  0 = not ready, 1 = AGGREGATE, 2 = INCLUSIVE
// written with a single agent-scope relaxed atomic store and polled with agent-scope relaxed loads (nothing else is
// handed off, so no fence is needed).  Carrying the tail means the output word shared by two tiles is written by
// exactly one of them (the later one): no zero-fill of the 2-bit stream, no atomics on it; only the sparse
// record-break bits use atomicOr on a zeroed bitmap.
#include <hip/hip_runtime.h>

#include "lash_kernels.h"

namespace lash {

// Four ASCII bytes -> four 2-bit codes (byte 0 first, in bits 7:6 of the result) and a 4-bit validity mask.
// code = (c >> 1) & 3 maps A C T G -> 0 1 2 3; x ^ (x >> 1) swaps 2 and 3 to get kmerutils' A C G T = 0 1 2 3.
// A byte is valid iff it equals the letter its own code would decode to (exact zero-byte test, no LUT).
// `tab4` (LayoutDev::code_tab4) maps those hypothesis codes to the context layout's, one v_perm per 4 bytes (0x03020100 = identity).
__device__ __forceinline__ void classify4(uint32_t w, uint32_t tab4, uint32_t &codes8, uint32_t &valid4)
{
    const uint32_t x = (w >> 1) & 0x03030303u;
    uint32_t e = (x << 1) | 0x41414141u;                   // 'A' 'C' 'E' 'G'
    const uint32_t m = (x >> 1) & ~x & 0x01010101u;        // 1 where x == 2
    e ^= m | (m << 4);                                     // 'E' ^ 0x11 = 'T'
    const uint32_t z = w ^ e;                              // zero byte <=> valid
    const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
    const uint32_t v = (~nz & 0x80808080u) >> 7;           // 0x01 per valid byte
    const uint32_t c = __builtin_amdgcn_perm(0u, tab4, x ^ ((x >> 1) & 0x01010101u));
    codes8 = (c * 0x40100401u) >> 24;                      // byte j -> bits 7-2j..6-2j
    valid4 = ((v * 0x01020408u) >> 24) & 0xFu;             // byte j -> bit j
}

struct Lane16 {
    uint32_t bits;    // surviving bases, first in bits 31:30
    uint32_t cnt;     // how many survive
    uint32_t vmask;   // bit j: byte j survives
};

__device__ __forceinline__ Lane16 classify16(const uint4 q, uint32_t keep, uint32_t tab4)
{
    uint32_t c0, c1, c2, c3, v0, v1, v2, v3;
    classify4(q.x, tab4, c0, v0);
    classify4(q.y, tab4, c1, v1);
    classify4(q.z, tab4, c2, v2);
    classify4(q.w, tab4, c3, v3);
    const uint32_t codes = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
    const uint32_t vmask = (v0 | (v1 << 4) | (v2 << 8) | (v3 << 12)) & keep;
    Lane16 o;
    o.vmask = vmask;
    if (vmask == 0xFFFFu) { o.bits = codes; o.cnt = 16; return o; }
    uint32_t bits = 0, cnt = 0, m = vmask;
    while (m) {                                             // rare path: stream compaction inside the lane
        const uint32_t j = (uint32_t)__builtin_ctz(m);
        m &= m - 1;
        bits |= ((codes >> (30 - 2 * j)) & 3u) << (30 - 2 * cnt);
        ++cnt;
    }
    o.bits = bits;
    o.cnt = cnt;
    return o;
}


constexpr int P2_THREADS = 256;
constexpr int P2_CHUNKS = 4;
constexpr int P2_TILE = P2_THREADS * P2_CHUNKS * 16;       // 16 KiB of one genome: 4 coalesced 16-byte chunks per lane
constexpr uint32_t P2_SPIN_LIMIT = 1u << 24;
constexpr uint32_t TF_FIRST = 1u, TF_LAST = 2u, TF_FULL = 4u, TF_FASTA = 8u, TF_FASTQ = 16u;

__device__ __forceinline__ uint64_t desc_make(uint32_t count, uint32_t tail30, uint32_t status)
{
    return ((uint64_t)count << 32) | ((uint64_t)(tail30 & 0x3FFFFFFFu) << 2) | status;
}
__device__ __forceinline__ uint32_t desc_count(uint64_t d) { return (uint32_t)(d >> 32); }
__device__ __forceinline__ uint32_t desc_tail(uint64_t d) { return ((uint32_t)d >> 2) & 0x3FFFFFFFu; }
// A (earlier) ++ B (later): on entry (cnt, tail) is B, on exit it is the concatenation
__device__ __forceinline__ void agg_combine(uint32_t &cnt, uint32_t &tail, uint32_t a_cnt, uint32_t a_tail)
{
    const uint32_t nb = cnt < 15u ? cnt : 15u;
    tail = (uint32_t)((((uint64_t)a_tail << (2 * nb)) | tail) & 0x3FFFFFFFu);
    cnt += a_cnt;
}

// ------------------------------------------------------------------------------------------------------------
// map kernel: one thread per tile -> TileInfo (which genome, where in seq, which records start inside)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_map_kernel(PackMapArgs m)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_tiles = m.n_tiles_dev ? *m.n_tiles_dev : m.n_tiles;
    if (t >= n_tiles) return;
    uint32_t lo = 0, hi = m.n_genomes;                       // last g with tile_begin[g] <= t (genomes without
    while (hi - lo > 1) {                                    // tiles share tile_begin with their successor)
        const uint32_t mid = (lo + hi) >> 1;
        if (m.tile_begin[mid] <= t) lo = mid; else hi = mid;
    }
    const uint32_t g = lo;
    const GenomeDesc gd = m.genomes[g];
    const uint32_t tb = m.tile_begin[g], te = m.tile_begin[g + 1];
    const int64_t g0 = (int64_t)gd.byte_off, g1 = g0 + (int64_t)gd.byte_len;
    const int64_t lead = (int64_t)((reinterpret_cast<uintptr_t>(m.seq) + gd.byte_off) & 15u);   // 16-B aligned loads
    const int64_t toff = g0 - lead + (int64_t)(t - tb) * P2_TILE;                                // relative to seq
    auto lower_bound = [&](int64_t x) {                      // first r in [rec_begin, rec_end] with rec_off[r] >= x
        if (gd.format != 0u) return (uint64_t)0;
        uint64_t a = gd.rec_begin, b = gd.rec_end;
        while (a < b) {
            const uint64_t mid = (a + b) >> 1;
            if ((int64_t)m.rec_off[mid] < x) a = mid + 1; else b = mid;
        }
        return a;
    };
    TileInfo ti;
    ti.toff = toff;
    ti.r0 = lower_bound(toff);
    const uint64_t r1 = (t + 1 < te) ? lower_bound(toff + P2_TILE) : gd.rec_end;
    ti.nrec = (uint32_t)(r1 - ti.r0 > 0xFFFFFFFFull ? 0xFFFFFFFFull : r1 - ti.r0);
    if (gd.format != 0u) { ti.r0 = 0; ti.nrec = 0; }        // raw file bytes: records are found on the device
    else if (gd.rec_end - gd.rec_begin <= 1) ti.nrec = 0;    // a single record has no interior boundary to mark
    ti.g = g;
    ti.tb = tb;
    ti.rel_lo = (int32_t)(g0 > toff ? g0 - toff : 0);
    ti.rel_hi = (int32_t)(g1 - toff < P2_TILE ? g1 - toff : P2_TILE);
    ti.flags = (t == tb ? TF_FIRST : 0u) | (t + 1 == te ? TF_LAST : 0u) |
               ((toff >= g0 && toff + P2_TILE <= g1) ? TF_FULL : 0u) |
               (gd.format == 1u ? TF_FASTA : gd.format == 2u ? TF_FASTQ : 0u);
    ti.pad = 0;
    ti.word_off = gd.word_off;
    ti.brk_off = gd.brk_off;
    m.tiles[t] = ti;
}

// TileInfo through the vector path, then pinned into SGPRs (the index is wave-uniform): keeps two tiles' worth of
// descriptors out of the VGPR budget
__device__ __forceinline__ TileInfo load_tile_uniform(const TileInfo *p)
{
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 x = q[i];
        w[4 * i + 0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.x);
        w[4 * i + 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.y);
        w[4 * i + 2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.z);
        w[4 * i + 3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.w);
    }
    TileInfo t;
    t.toff = (int64_t)(((uint64_t)w[1] << 32) | w[0]);
    t.r0 = ((uint64_t)w[3] << 32) | w[2];
    t.nrec = w[4];
    t.g = w[5];
    t.tb = w[6];
    t.rel_lo = (int32_t)w[7];
    t.rel_hi = (int32_t)w[8];
    t.flags = w[9];
    t.pad = 0;
    t.word_off = ((uint64_t)w[13] << 32) | w[12];
    t.brk_off = ((uint64_t)w[15] << 32) | w[14];
    return t;
}

__device__ __forceinline__ uint4 load16_clipped(const uint8_t *seq, int64_t off, int64_t seq_bytes)
{
    // 16 bytes at seq[off .. off+16), bytes outside [0, seq_bytes) read as 0 (only the batch's first / last chunk)
    if (off >= 0 && off + 16 <= seq_bytes) return *reinterpret_cast<const uint4 *>(seq + off);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int j = 0; j < 16; ++j) {
        const int64_t o = off + j;
        if (o >= 0 && o < seq_bytes) w[j >> 2] |= (uint32_t)seq[o] << (8 * (j & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------------------------------------------------
// the pack kernel: persistent workgroups, software-pipelined over tiles
//
//   iteration i:  classify(T_i) -> loads(T_i+1) go out -> stage(T_i), publish AGGREGATE(T_i)
//                 -> look-back + publish INCLUSIVE(T_i-1) -> store(T_i-1)
//
// * a tile's AGGREGATE is published as soon as its bytes are classified, and its own look-back runs one iteration
//   later, when its predecessors' aggregates are (almost always) already there: little spinning, and a drawn tile
//   is never left idle where successors would wait on it;
// * the next tile's 16 KiB of loads are in flight during the whole second half of the iteration.  Workgroup barriers
//   are raw s_barrier + s_waitcnt lgkmcnt(0) (LDS only): __syncthreads() would also drain vmcnt, i.e. the prefetch.
//
// Tickets are sharded: workgroup b serves shard b % S, and shard s hands out tiles s, s+S, s+2S, ... in order from
// its own counter (one L2 atomic word saturates near 88 draws/us; 16 KiB tiles need ~300/us at HBM speed).
// Deadlock-free for ANY residency: let m be the smallest tile whose AGGREGATE is not published.  Tiles < m all have
// aggregates, so every look-back below m terminates; m's holder therefore finishes what it does before reaching
// m (only tiles < m), and if m is not drawn yet its shard's workgroups hold only smaller tiles.
// ------------------------------------------------------------------------------------------------------------
// ---- raw FASTA / FASTQ bytes (the parse moves onto the device) -------------------------------
// needletail semantics: FASTA = '>' header line, then sequence lines up to the next '>';
// FASTQ = 4-line records, line 2 is the sequence.  Newlines, '\r' and everything else that is not ACGT are dropped by
// the same filter that implements filter_out_n; what is added here is (a) header / '+' / quality lines are dropped
// even where they contain ACGT, (b) a record starts at every '>' that opens a line (FASTA) or header-ending newline
// (FASTQ).
// Line state is a prefix property of the file: "last of {'>' -> header, '\n' -> sequence}" for FASTA, "newlines so far
// mod 4" for FASTQ.  Inside a tile it is resolved with ballots / shuffles, across tiles with a second, 4-byte
// look-back descriptor (bits 1:0 status, FASTA: bit 2 = tile has a setter, bit 3 = state after it; FASTQ: bits 3:2 =
// newline count mod 4 / phase at the tile's end).
__device__ __forceinline__ uint32_t eqmask16(const uint4 q, uint32_t pat4)
{
    auto m4 = [&](uint32_t w) {
        const uint32_t z = w ^ pat4;
        const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
        const uint32_t v = (~nz & 0x80808080u) >> 7;
        return ((v * 0x01020408u) >> 24) & 0xFu;
    };
    return m4(q.x) | (m4(q.y) << 4) | (m4(q.z) << 8) | (m4(q.w) << 12);
}

// one lane's 16 bytes of FASTA: which are inside a header line, which '>' start a record
__device__ __forceinline__ void fasta_chunk(uint32_t gt, uint32_t nl, uint32_t in_hdr, uint32_t &hdrmask, uint32_t &recstart)
{
    uint32_t S = gt | nl, state = in_hdr, prev = 0;
    hdrmask = 0;
    recstart = 0;
    while (S) {
        const uint32_t p = (uint32_t)__builtin_ctz(S);
        S &= S - 1;
        if (state) hdrmask |= ((1u << p) - 1u) & ~((1u << prev) - 1u);
        const uint32_t is_gt = (gt >> p) & 1u;
        if (is_gt && !state) recstart |= 1u << p;
        state = is_gt;
        prev = p;
    }
    if (state) hdrmask |= 0xFFFFu & ~((1u << prev) - 1u);
}

// one lane's 16 bytes of FASTQ: which belong to a sequence line, which newlines end a header line
__device__ __forceinline__ void fastq_chunk(uint32_t nl, uint32_t in_phase, uint32_t &seqmask, uint32_t &recstart)
{
    uint32_t S = nl, ph = in_phase, prev = 0;
    seqmask = 0;
    recstart = 0;
    while (S) {
        const uint32_t p = (uint32_t)__builtin_ctz(S);
        S &= S - 1;
        if (ph == 1u) seqmask |= ((1u << p) - 1u) & ~((1u << prev) - 1u);
        if (ph == 0u) recstart |= 1u << p;
        ph = (ph + 1u) & 3u;
        prev = p + 1u;
    }
    if (ph == 1u) seqmask |= 0xFFFFu & ~((1u << prev) - 1u);
}

// FASTQ structure check (needletail stops iterating at a record that is not header / sequence / '+' / quality, and lash keeps
// what came before: utils.rs:457): a line in phase 0 must start with '@', one in phase 2 with '+'.  Returns true when one of
// this lane's 16 bytes starts such a line with another character.  (Quality lines may start with anything, '@' included:
// that is why the phase, not the character, says which line this is.)
__device__ __forceinline__ bool fastq_bad_line_start(uint32_t nl, uint32_t at, uint32_t plus, uint32_t keep, uint32_t in_phase,
                                                     int start_pos)
{
    uint32_t ph = in_phase, bad = 0;
    if (start_pos >= 0 && start_pos < 16 && ((keep >> start_pos) & 1u))
        bad |= (ph == 0u && !((at >> start_pos) & 1u)) || (ph == 2u && !((plus >> start_pos) & 1u));
    uint32_t S = nl;
    while (S) {
        const uint32_t nx = (uint32_t)__builtin_ctz(S) + 1u;
        S &= S - 1;
        ph = (ph + 1u) & 3u;
        if (nx < 16u && ((keep >> nx) & 1u)) bad |= (ph == 0u && !((at >> nx) & 1u)) || (ph == 2u && !((plus >> nx) & 1u));
    }
    return bad != 0u;
}

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct CarriedTile {          // what iteration i+1 needs to finish tile T_i (all wave-uniform)
    uint64_t word_off, brk_off;
    uint32_t t, tb, g, flags, tile_cnt, own_tail, has_rec, valid;
};

// RAW = false: every genome comes as record sequences + rec_off (lean kernel); RAW = true: some genomes are raw file bytes
template <bool RAW>
__global__ void __launch_bounds__(P2_THREADS) pack_lookback_kernel(PackArgs a, PackV2Args v)
{
    constexpr int STAGE_WORDS = P2_TILE / 16 + 8;
    __shared__ uint32_t stage[2][STAGE_WORDS];              // surviving bases of tile i (buffer i & 1), from bit 31 of word 0
    __shared__ uint32_t recbits[P2_TILE / 32];              // which bytes of the current tile start a record
    __shared__ uint32_t brkloc[2][P2_TILE / 32];            // the same in tile-local compacted positions
    __shared__ uint32_t row_tot[P2_CHUNKS][P2_THREADS / 64];
    __shared__ uint32_t wave_flag[P2_THREADS / 64];
    __shared__ uint32_t grp[P2_CHUNKS * (P2_THREADS / 64)];  // raw mode: per (chunk row, wave) line-state summary
    __shared__ uint32_t s_next, s_prev_cnt, s_prev_tail, s_fail, s_own_tail, s_line_in, s_rawfail;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t seq_bytes = a.seq_end - a.seq;
    const uint32_t shard = blockIdx.x % v.n_shards;
    uint32_t *const my_ticket = v.ticket + shard * 32u;     // 128-byte stride
    const uint32_t n_tiles = v.n_tiles_dev ? *v.n_tiles_dev : v.n_tiles;

    auto issue_loads = [&](const TileInfo &ti, uint4 (&q)[P2_CHUNKS]) {
        if (ti.flags & TF_FULL) {
#pragma unroll
            for (int c = 0; c < P2_CHUNKS; ++c)
                q[c] = *reinterpret_cast<const uint4 *>(a.seq + ti.toff + (int64_t)((c * P2_THREADS + (int)tid) * 16));
        } else {
#pragma unroll
            for (int c = 0; c < P2_CHUNKS; ++c) {
                const int32_t cs = (c * P2_THREADS + (int)tid) * 16;
                q[c] = (cs + 16 > ti.rel_lo && cs < ti.rel_hi) ? load16_clipped(a.seq, ti.toff + cs, seq_bytes)
                                                                : make_uint4(0, 0, 0, 0);
            }
        }
    };

    // ---- prologue: first tile, its loads, and the ticket after it ----
    if (tid == 0) { s_next = shard + v.n_shards * atomicAdd(my_ticket, 1u); s_rawfail = 0; }
    lds_barrier();
    uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_next);
    if (t >= n_tiles) return;
    TileInfo ti = load_tile_uniform(v.tiles + t);
    uint4 q[P2_CHUNKS];
    issue_loads(ti, q);
    lds_barrier();                                          // everyone has read s_next
    if (tid == 0) s_next = shard + v.n_shards * atomicAdd(my_ticket, 1u);

    CarriedTile ct{};
    bool have_cur = true;
    for (uint32_t it = 0;; ++it) {
        const uint32_t buf = it & 1u;
        uint32_t excl[P2_CHUNKS], tile_cnt = 0;
        Lane16 l16[P2_CHUNKS];
        const bool raw = RAW && have_cur && (ti.flags & (TF_FASTA | TF_FASTQ)) != 0;          // uniform
        const bool has_rec = have_cur && (ti.nrec != 0 || raw);                                // uniform
        bool have_next = false;
        uint32_t t_next = 0xFFFFFFFFu;
        TileInfo ti_next = ti;

        if (have_cur) {
            // ---- 1. record starts inside this tile (records mode: from rec_off, sparse; raw mode: found below) ----
            if (has_rec) {
                for (uint32_t i = tid; i < P2_TILE / 32; i += P2_THREADS) { recbits[i] = 0; brkloc[buf][i] = 0; }
                lds_barrier();
                for (uint64_t r = ti.r0 + tid; r < ti.r0 + ti.nrec; r += P2_THREADS) {
                    const int64_t off = (int64_t)a.rec_off[r] - ti.toff;
                    if (off >= 0 && off < P2_TILE) atomicOr(&recbits[off >> 5], 1u << (off & 31));
                }
            }
            // ---- 2. which bytes belong to the genome; raw mode: which of those are sequence ----
            uint32_t keep[P2_CHUNKS];
#pragma unroll
            for (int c = 0; c < P2_CHUNKS; ++c) {
                keep[c] = 0xFFFFu;
                if (!(ti.flags & TF_FULL)) {
                    const int32_t cs = (c * P2_THREADS + (int)tid) * 16;
                    int32_t lo = ti.rel_lo - cs, hi = ti.rel_hi - cs;
                    lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
                    hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
                    keep[c] = hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
                }
            }
            if constexpr (RAW) if (raw) {
                const bool fasta = (ti.flags & TF_FASTA) != 0;                                 // uniform
                uint32_t nl[P2_CHUNKS], gt[P2_CHUNKS], lane_in[P2_CHUNKS];
                uint32_t at_m[P2_CHUNKS], plus_m[P2_CHUNKS], q_last[P2_CHUNKS];   // FASTQ structure check (a.file_err)
                uint32_t unresolved = 0;                       // bit c: lane_in[c] still needs the group / tile state
#pragma unroll
                for (int c = 0; c < P2_CHUNKS; ++c) {
                    nl[c] = eqmask16(q[c], 0x0A0A0A0Au) & keep[c];
                    if (!fasta && a.file_err) {
                        at_m[c] = eqmask16(q[c], 0x40404040u);
                        plus_m[c] = eqmask16(q[c], 0x2B2B2B2Bu);
                        q_last[c] = q[c].w >> 24;
                    } else { at_m[c] = 0; plus_m[c] = 0; q_last[c] = 0; }
                    uint32_t g = fasta ? (eqmask16(q[c], 0x3E3E3E3Eu) & keep[c]) : 0u;
                    if (g) {
                        // needletail finds the next record at "\n>": a '>' in the middle of a line is sequence text (and is
                        // then deleted by the ACGT filter).  Bytes 1..15 look at their left neighbour in the chunk; byte 0
                        // at the byte before the chunk (rare: one global byte load), or at nothing when it opens the file.
                        uint32_t after_nl = (nl[c] << 1) & 0xFFFFu;
                        const int32_t cs = (c * P2_THREADS + (int)tid) * 16;
                        const int32_t first = (ti.flags & TF_FIRST) ? ti.rel_lo - cs : -1;     // the file's first byte, if in this chunk
                        if (first >= 0 && first < 16) after_nl |= 1u << first;
                        else if ((g & 1u) && a.seq[ti.toff + cs - 1] == (uint8_t)'\n') after_nl |= 1u;
                        g &= after_nl;
                    }
                    gt[c] = g;
                }
                if (fasta) {
#pragma unroll
                    for (int c = 0; c < P2_CHUNKS; ++c) {
                        const uint32_t S = gt[c] | nl[c];
                        const bool has = S != 0u;
                        const bool end_hdr = has && ((gt[c] >> (31 - __builtin_clz(S | 1u))) & 1u);
                        const uint64_t m_has = __builtin_amdgcn_ballot_w64(has), m_hdr = __builtin_amdgcn_ballot_w64(end_hdr);
                        const uint64_t prior = m_has & ((1ull << lane) - 1ull);
                        if (prior) lane_in[c] = (uint32_t)(m_hdr >> (63 - __builtin_clzll(prior))) & 1u;
                        else { lane_in[c] = 0; unresolved |= 1u << c; }
                        if (lane == 0)
                            grp[c * (P2_THREADS / 64) + wid] = m_has ? (1u | ((uint32_t)((m_hdr >> (63 - __builtin_clzll(m_has))) & 1u) << 1)) : 0u;
                    }
                } else {
                    uint32_t inc[P2_CHUNKS];
#pragma unroll
                    for (int c = 0; c < P2_CHUNKS; ++c) inc[c] = (uint32_t)__builtin_popcount(nl[c]);
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
                        for (int c = 0; c < P2_CHUNKS; ++c) {
                            const uint32_t n = __shfl_up(inc[c], d, 64);
                            if (lane >= (uint32_t)d) inc[c] += n;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < P2_CHUNKS; ++c) {
                        lane_in[c] = inc[c] - (uint32_t)__builtin_popcount(nl[c]);            // newlines before, in this wave row
                        if (lane == 63) grp[c * (P2_THREADS / 64) + wid] = inc[c];
                    }
                    unresolved = 0xFu;
                }
                lds_barrier();
                // groups in tile order g = c * 4 + w; wave 0 also resolves the tile's incoming state by look-back
                constexpr int NG = P2_CHUNKS * (P2_THREADS / 64);
                if (wid == 0) {
                    uint32_t agg, fail = 0, in_state = 0;
                    if (fasta) {
                        agg = 0;
                        for (int g = NG - 1; g >= 0; --g) { const uint32_t x = grp[g]; if (x & 1u) { agg = 1u | (x & 2u); break; } }
                    } else {
                        uint32_t tot = 0;
                        for (int g = 0; g < NG; ++g) tot += grp[g];
                        agg = tot & 3u;
                    }
                    // desc2 word: status | payload << 2   (FASTA payload: has | state << 1; FASTQ: count / phase)
                    if (!(ti.flags & TF_FIRST)) {
                        if (lane == 0) __hip_atomic_store(v.desc2 + t, 1u | (agg << 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        int64_t j = (int64_t)t - 1;
                        uint32_t spins = 0, acc = 0;
                        for (;;) {
                            const int64_t idx = j - (int64_t)lane;
                            uint32_t d = 2u;                                   // before the file's first tile: state 0
                            if (idx >= (int64_t)ti.tb) d = __hip_atomic_load(v.desc2 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const uint32_t st = d & 3u, pay = d >> 2;
                            const bool decisive = st == 2u || (fasta && st == 1u && (pay & 1u));
                            const uint64_t m_dec = __builtin_amdgcn_ballot_w64(decisive);
                            const uint64_t m_none = __builtin_amdgcn_ballot_w64(st == 0u);
                            const int first = m_dec ? __builtin_ctzll(m_dec) : 64;
                            const uint64_t need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
                            if (m_none & need) {
                                if (++spins > P2_SPIN_LIMIT) { fail = 1; break; }
                                __builtin_amdgcn_s_sleep(2);
                                continue;
                            }
                            if (fasta) {
                                if (first < 64) { in_state = (uint32_t)__builtin_amdgcn_readlane((int)(pay >> 1), first) & 1u; break; }
                            } else {
                                // phase = inclusive phase of lane `first` + newline counts of the nearer aggregates
                                const uint64_t below = first < 64 ? need : ~0ull;
                                const uint64_t b0 = __builtin_amdgcn_ballot_w64((pay & 1u) != 0) & below;
                                const uint64_t b1 = __builtin_amdgcn_ballot_w64((pay & 2u) != 0) & below;
                                acc += (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1);
                                if (first < 64) { in_state = acc & 3u; break; }
                            }
                            j -= 64;
                        }
                    }
                    const uint32_t end_state = fasta ? ((agg & 1u) ? (agg >> 1) & 1u : in_state) : (in_state + agg) & 3u;
                    if (lane == 0) {
                        if (!fail) __hip_atomic_store(v.desc2 + t, 2u | ((fasta ? (1u | (end_state << 1)) : end_state) << 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_line_in = in_state;
                        if (fail) s_rawfail = 1;
                    }
                }
                lds_barrier();
                if (s_rawfail) { if (tid == 0) atomicOr(v.error_flag, 1u); return; }
                const uint32_t tile_in = s_line_in;
                bool rs_any = false;
#pragma unroll
                for (int c = 0; c < P2_CHUNKS; ++c) {
                    const int g = c * (P2_THREADS / 64) + (int)wid;
                    uint32_t in = lane_in[c];
                    if (fasta) {
                        if (unresolved & (1u << c)) {
                            in = tile_in;
                            for (int k = g - 1; k >= 0; --k) { const uint32_t x = grp[k]; if (x & 1u) { in = (x >> 1) & 1u; break; } }
                        }
                        uint32_t hdrmask, rs;
                        fasta_chunk(gt[c], nl[c], in, hdrmask, rs);
                        keep[c] &= ~hdrmask;
                        lane_in[c] = rs;
                    } else {
                        uint32_t before = tile_in;
                        for (int k = 0; k < g; ++k) before += grp[k];
                        if (a.file_err) {
                            // which byte of this chunk, if any, starts a line whose phase is the chunk's incoming phase: the
                            // file's first byte (first tile: offset rel_lo of chunk 0), else byte 0 when a newline precedes it
                            // (the lane before holds that byte; a wave's first lane reads it from memory)
                            const int32_t cs = (c * P2_THREADS + (int)tid) * 16;
                            uint32_t prev = __shfl_up(q_last[c], 1, 64);
                            int start_pos = -1;
                            if ((ti.flags & TF_FIRST) && cs == 0) start_pos = ti.rel_lo;
                            else if (cs < ti.rel_hi) {
                                if (lane == 0) prev = a.seq[ti.toff + cs - 1];
                                if (prev == 0x0Au) start_pos = 0;
                            }
                            if (fastq_bad_line_start(nl[c], at_m[c], plus_m[c], keep[c], (in + before) & 3u, start_pos))
                                atomicOr(a.file_err + ti.g, 1u);
                        }
                        uint32_t seqmask, rs;
                        fastq_chunk(nl[c], (in + before) & 3u, seqmask, rs);
                        keep[c] &= seqmask;
                        lane_in[c] = rs;
                    }
                    rs_any = rs_any || lane_in[c] != 0u;
                }
                // record-start marks into the tile bitmap (two lanes share a word)
                if (rs_any) {
#pragma unroll
                    for (int c = 0; c < P2_CHUNKS; ++c) {
                        const uint32_t ci = (uint32_t)(c * P2_THREADS) + tid;
                        if (lane_in[c]) atomicOr(&recbits[ci >> 1], lane_in[c] << ((ci & 1u) * 16u));
                    }
                }
            }
            // ---- 2b. classify (records mode: waits for this tile's loads here) ----
            bool allv = true;
#pragma unroll
            for (int c = 0; c < P2_CHUNKS; ++c) {
                l16[c] = keep[c] ? classify16(q[c], keep[c], a.code_tab4) : Lane16{0, 0, 0};
                allv = allv && l16[c].vmask == 0xFFFFu;
            }
            const bool wave_all = __builtin_amdgcn_ballot_w64(!allv) == 0ull;
            if (lane == 0) wave_flag[wid] = wave_all ? 1u : 0u;
        }
        lds_barrier();                                      // B2: wave_flag, s_next, recbits visible
        bool clean = false;
        if (have_cur) {
            clean = (wave_flag[0] & wave_flag[1] & wave_flag[2] & wave_flag[3]) != 0;
            // ---- 3. next tile: its descriptor and its loads go out now and stay in flight ----
            t_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_next);
            have_next = t_next < n_tiles;
            if (have_next) {
                ti_next = load_tile_uniform(v.tiles + t_next);
                issue_loads(ti_next, q);
            }
            // ---- 4. positions inside the tile + staging of the surviving bases into stage[buf] ----
            if (clean) {
                // nothing was deleted: lane's chunk c is exactly word (c*256 + tid) of the tile-local stream
                tile_cnt = P2_TILE;
#pragma unroll
                for (int c = 0; c < P2_CHUNKS; ++c) {
                    const uint32_t ci = (uint32_t)(c * P2_THREADS) + tid;
                    excl[c] = ci * 16u;
                    stage[buf][ci] = l16[c].bits;
                }
                if (tid < 8) stage[buf][P2_TILE / 16 + tid] = 0;
            } else {
                uint32_t inc[P2_CHUNKS];
#pragma unroll
                for (int c = 0; c < P2_CHUNKS; ++c) inc[c] = l16[c].cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
                    for (int c = 0; c < P2_CHUNKS; ++c) {
                        const uint32_t n = __shfl_up(inc[c], d, 64);
                        if (lane >= (uint32_t)d) inc[c] += n;
                    }
                }
                if (lane == 63) {
#pragma unroll
                    for (int c = 0; c < P2_CHUNKS; ++c) row_tot[c][wid] = inc[c];
                }
                for (uint32_t i = tid; i < (uint32_t)STAGE_WORDS; i += P2_THREADS) stage[buf][i] = 0;
                lds_barrier();
                tile_cnt = 0;
#pragma unroll
                for (int c = 0; c < P2_CHUNKS; ++c) {
                    uint32_t base = tile_cnt;
#pragma unroll
                    for (int w = 0; w < P2_THREADS / 64; ++w) {
                        const uint32_t x = row_tot[c][w];
                        if (w < (int)wid) base += x;
                        tile_cnt += x;
                    }
                    excl[c] = base + inc[c] - l16[c].cnt;
                }
#pragma unroll
                for (int c = 0; c < P2_CHUNKS; ++c) {
                    if (l16[c].cnt) {
                        const uint32_t wi = excl[c] >> 4, sh = (excl[c] & 15u) * 2u;
                        atomicOr(&stage[buf][wi], l16[c].bits >> sh);
                        if (sh) atomicOr(&stage[buf][wi + 1], l16[c].bits << (32 - sh));
                    }
                }
            }
            // record starts -> tile-local compacted positions (the global offset is known one iteration later)
            if (has_rec) {
#pragma unroll
                for (int c = 0; c < P2_CHUNKS; ++c) {
                    const uint32_t ci = (uint32_t)(c * P2_THREADS) + tid;
                    uint32_t m = (recbits[ci >> 1] >> ((ci & 1u) * 16u)) & 0xFFFFu;
                    while (m) {
                        const uint32_t jb = (uint32_t)__builtin_ctz(m);
                        m &= m - 1;
                        const uint32_t pos = excl[c] + (uint32_t)__builtin_popcount(l16[c].vmask & ((1u << jb) - 1u));
                        atomicOr(&brkloc[buf][pos >> 5], 1u << (pos & 31));      // pos <= byte offset < P2_TILE
                    }
                }
            }
        }
        lds_barrier();                                      // B3: stage[buf] complete

        // ---- 5. wave 0: publish AGGREGATE(cur); draw a ticket; look back for the carried tile; publish its INCLUSIVE ----
        if (wid == 0) {
            uint32_t nn = 0xFFFFFFFFu;
            uint32_t own_tail = 0;
            if (have_cur) {
                if (lane == 0 && have_next) nn = shard + v.n_shards * atomicAdd(my_ticket, 1u);
                const uint32_t nb = tile_cnt < 15u ? tile_cnt : 15u;    // last min(cnt,15) staged bases, right-aligned
                if (nb) {
                    const uint32_t wi = (tile_cnt - 1) >> 4;
                    const uint64_t win = ((uint64_t)(wi ? stage[buf][wi - 1] : 0u) << 32) | stage[buf][wi];
                    const uint32_t used = ((tile_cnt - 1) & 15u) + 1u;   // bases of word wi in use
                    own_tail = (uint32_t)((win >> (32 - 2 * used)) & ((1ull << (2 * nb)) - 1ull));
                }
                if (lane == 0)
                    __hip_atomic_store(v.desc + t, desc_make(tile_cnt, own_tail, (ti.flags & TF_FIRST) ? 2u : 1u),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint32_t p_cnt = 0, p_tail = 0, fail = 0;
            if (ct.valid && !(ct.flags & TF_FIRST)) {
                int64_t j = (int64_t)ct.t - 1;              // window of 64 predecessors: lane i looks at tile j - i
                uint32_t spins = 0;
                for (;;) {
                    const int64_t idx = j - (int64_t)lane;
                    uint64_t d = desc_make(0, 0, 2u);       // before the genome's first tile: the empty prefix
                    if (idx >= (int64_t)ct.tb) d = __hip_atomic_load(v.desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t st = (uint32_t)d & 3u;
                    const uint64_t m_incl = __builtin_amdgcn_ballot_w64(st == 2u);
                    const uint64_t m_none = __builtin_amdgcn_ballot_w64(st == 0u);
                    const int first_incl = m_incl ? __builtin_ctzll(m_incl) : 64;
                    const uint64_t need = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);
                    if (m_none & need) {                    // a needed predecessor has not published yet
                        if (++spins > P2_SPIN_LIMIT) { fail = 1; break; }
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                    // ordered product D[first_incl] ++ ... ++ D[0] by a 6-step shuffle tree: lane i ends with
                    // D[i+2^k-1] ++ ... ++ D[i]; lanes past the first INCLUSIVE one are the identity (0, 0)
                    uint32_t w_cnt = (int)lane <= first_incl ? desc_count(d) : 0u;
                    uint32_t w_tail = (int)lane <= first_incl ? desc_tail(d) : 0u;
#pragma unroll
                    for (int sft = 1; sft < 64; sft <<= 1) {
                        uint32_t e_cnt = __shfl_down(w_cnt, sft, 64), e_tail = __shfl_down(w_tail, sft, 64);
                        if (lane + (uint32_t)sft >= 64u) { e_cnt = 0; e_tail = 0; }
                        agg_combine(w_cnt, w_tail, e_cnt, e_tail);           // lanes i+sft.. are EARLIER than lane i
                    }
                    w_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_cnt);
                    w_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_tail);
                    agg_combine(p_cnt, p_tail, w_cnt, w_tail);               // the window is earlier than p
                    if (first_incl < 64) break;
                    j -= 64;
                }
                uint32_t i_cnt = ct.tile_cnt, i_tail = ct.own_tail;
                agg_combine(i_cnt, i_tail, p_cnt, p_tail);
                if (lane == 0 && !fail)
                    __hip_atomic_store(v.desc + ct.t, desc_make(i_cnt, i_tail, 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) { s_prev_cnt = p_cnt; s_prev_tail = p_tail; s_fail = fail; s_next = nn; s_own_tail = own_tail; }
        }
        lds_barrier();                                      // B4
        if (s_fail) { if (tid == 0) atomicOr(v.error_flag, 1u); return; }

        // ---- 6. finish the carried tile: break bits and output words (its bases are in stage[buf ^ 1]) ----
        if (ct.valid) {
            const uint32_t pb = buf ^ 1u;
            const uint32_t prev = s_prev_cnt;
            if (ct.has_rec) {
                for (uint32_t i = tid; i < P2_TILE / 32; i += P2_THREADS) {
                    uint32_t m = brkloc[pb][i];
                    while (m) {
                        const uint32_t jb = (uint32_t)__builtin_ctz(m);
                        m &= m - 1;
                        const uint64_t pos = (uint64_t)prev + i * 32u + jb;
                        atomicOr(a.brk + ct.brk_off + (pos >> 5), 1u << (pos & 31));
                    }
                }
            }
            const uint32_t carry = prev & 15u;
            const uint32_t cw = s_prev_tail & (carry ? ((1u << (2 * carry)) - 1u) : 0u);     // right-aligned carried bases
            const uint32_t total = carry + ct.tile_cnt;
            const bool last_tile = (ct.flags & TF_LAST) != 0;
            const uint32_t nout = last_tile ? (total + 15u) >> 4 : total >> 4;
            uint32_t *dst = a.words + ct.word_off + (prev >> 4);
            for (uint32_t jw = tid; jw < nout; jw += P2_THREADS) {
                const uint32_t hiw = jw ? stage[pb][jw - 1] : cw;
                const uint32_t low = stage[pb][jw];
                dst[jw] = carry ? __builtin_amdgcn_alignbit(hiw, low, 2 * carry) : low;
            }
            if (last_tile) {
                if (tid == 0) a.nvalid[ct.g] = (uint64_t)prev + ct.tile_cnt;
                if (tid < PAD_WORDS) dst[nout + tid] = 0;      // defined look-ahead words for the sketch kernel
            }
        }
        if (!have_cur) return;                              // that was the drain iteration

        // ---- 7. rotate: cur becomes the carried tile ----
        ct.word_off = ti.word_off; ct.brk_off = ti.brk_off; ct.t = t; ct.tb = ti.tb; ct.g = ti.g; ct.flags = ti.flags;
        ct.tile_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_cnt);
        ct.own_tail = s_own_tail;
        ct.has_rec = has_rec ? 1u : 0u;
        ct.valid = 1u;
        have_cur = have_next;
        t = t_next;
        ti = ti_next;
        lds_barrier();                                      // B5: s_* / recbits are rewritten by the next iteration
    }
}

uint32_t pack_v2_tile_bytes() { return P2_TILE; }

// exclusive scan of the dirty genomes' tile counts (see launch_dirty_tile_scan in lash_kernels.h)
__global__ void __launch_bounds__(1024) dirty_tile_scan_kernel(const uint32_t *tile_begin, const uint32_t *dirty,
                                                               uint32_t n_genomes, uint32_t *tile_begin_c, uint32_t *n_tiles_c)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_genomes; base += 1024) {
        const uint32_t g = base + tid;
        const uint32_t x = (g < n_genomes && dirty[g]) ? tile_begin[g + 1] - tile_begin[g] : 0u;
        uint32_t inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t n = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += n;
        }
        if (lane == 63) wave_tot[wid] = inc;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wid; ++w) before += wave_tot[w];
        if (g < n_genomes) tile_begin_c[g] = before + inc - x;
        __syncthreads();
        if (tid == 1023) carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) { tile_begin_c[n_genomes] = carry; *n_tiles_c = carry; }
}

hipError_t launch_dirty_tile_scan(const uint32_t *tile_begin, const uint32_t *dirty, uint32_t n_genomes,
                                  uint32_t *tile_begin_c, uint32_t *n_tiles_c, hipStream_t stream)
{
    hipLaunchKernelGGL(dirty_tile_scan_kernel, dim3(1), dim3(1024), 0, stream, tile_begin, dirty, n_genomes, tile_begin_c, n_tiles_c);
    return hipGetLastError();
}

hipError_t launch_pack_v2(const PackArgs &args, const PackV2Args &v, const PackMapArgs &m, uint32_t cu_count, bool any_raw,
                          hipStream_t stream)
{
    if (v.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_map_kernel, dim3((v.n_tiles + 255) / 256), dim3(256), 0, stream, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    uint32_t grid = cu_count * 8u;                             // persistent workgroups, up to 8 per CU
    if (grid > v.n_tiles) grid = v.n_tiles;
    PackV2Args vv = v;
    vv.n_shards = grid < PACK_TICKET_SHARDS ? grid : PACK_TICKET_SHARDS;
    if (any_raw) hipLaunchKernelGGL(pack_lookback_kernel<true>, dim3(grid), dim3(P2_THREADS), 0, stream, args, vv);
    else hipLaunchKernelGGL(pack_lookback_kernel<false>, dim3(grid), dim3(P2_THREADS), 0, stream, args, vv);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// synthetic genomes: base i of genome g = bits 2*(i%32) of splitmix64((SEED ^ g*GOLDEN) + i/32)  
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) synth_kernel(uint64_t first_genome, uint64_t n_genomes, uint64_t n_bases,
                                                    uint64_t words_per_genome, uint8_t *out)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_genomes * words_per_genome) return;
    const uint64_t g = idx / words_per_genome, j = idx % words_per_genome;
    const uint64_t w = splitmix64((20260128ULL ^ ((first_genome + g) * 0x9E3779B97F4A7C15ULL)) + j);
    uint8_t *dst = out + g * n_bases + 32 * j;
    const uint64_t left = n_bases - 32 * j;
    uint32_t o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t c = (uint32_t)(w >> (2 * (4 * q + b))) & 3u;
            v |= ((0x54474341u >> (8 * c)) & 0xFFu) << (8 * b);            // "ACGT"[c]
        }
        o[q] = v;
    }
    if (left >= 32 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        reinterpret_cast<uint4 *>(dst)[0] = make_uint4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<uint4 *>(dst)[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        const uint64_t n = left < 32 ? left : 32;
        for (uint64_t b = 0; b < n; ++b) dst[b] = (uint8_t)(o[b >> 2] >> (8 * (b & 3)));
    }
}

hipError_t launch_synth(uint64_t first_genome, uint32_t n_genomes, uint64_t n_bases, uint8_t *d_out, hipStream_t stream)
{
    if (n_genomes == 0 || n_bases == 0) return hipSuccess;
    const uint64_t wpg = (n_bases + 31) / 32, total = wpg * n_genomes;
    const uint64_t blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, first_genome, (uint64_t)n_genomes,
                       n_bases, wpg, d_out);
    return hipGetLastError();
}

}  // namespace lash
