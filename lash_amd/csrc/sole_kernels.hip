// sole_kernels.hip — whole small genomes on persistent workgroups (round 5).
//
// The reference makes one sketch per FILE whatever its size (/root/reference/src/utils.rs:450-509): a collection of viruses,
// plasmids, amplicons or assembled contigs is very many genomes of a few kbp each.  sketch_kernel (sketch_kernels.hip) gives such
// a genome a workgroup of its own and pays, per genome, three dependent loads to find its bytes, a tile shape (64 bytes per lane)
// that a 10 kbp genome fills with 2.5 of 8 waves, and the flush: ~20 us of a workgroup slot against 1..7 us of hashing
// (profiles/r04/small_genomes_ablation.txt).  Here workgroups are RESIDENT and genomes stream through them:
//
//   * work = chunks of consecutive genomes of about equal cost, cut on the host from the genome byte offsets alone (no GenomeDesc,
//     no work items: the per-call host work is one pass over n + 1 offsets); a workgroup starts with chunk blockIdx.x and takes
//     further ones through a ticket counter (one atomic per chunk);
//   * a ROUND is 16 bytes per lane.  Every byte is converted to its 2-bit code exactly once, deleted bytes (filter_out_n,
//     utils.rs:33-41) are dropped on the way — survivors counted per lane, a DPP prefix sum per wave, the waves' totals through
//     LDS — and the survivors are appended to a ring of packed bases in LDS, record starts as bits of a second ring at the
//     position of the first surviving base at or after them (= the number of survivors before the record's first byte: no carry
//     logic, however many bytes are deleted in between).  The ring holds the genome as KSeq::new would see it (utils.rs:464);
//   * every lane then hashes ONE ring word — 16 k-mer starts, the clean path's process_word — of the words that were complete a
//     round earlier, so a round costs one workgroup barrier; a 10 kbp genome keeps all eight waves of a HyperMinHash workgroup busy
//     (one-wave workgroups for HyperLogLog / UltraLogLog tables of a few KiB: a wave per genome, no barrier at all);
//   * the loads of the NEXT round are issued before this one is hashed, and when this round is the genome's last they are the
//     first bytes of the next genome (which begins where this one ends): no genome waits for its descriptor or its first bytes;
//   * the image leaves LDS in 16-byte stores by all waves, the table is re-armed in the same sweep.
//
// Bit-exact by the same argument as everywhere else: max / OR are commutative and idempotent, so the order in which a genome's
// k-mers reach the table does not matter, and the k-mer multiset is the reference's: windows of the filtered
// sequence that do not span a record start (utils.rs:457-499).
#include <atomic>

#include "sketch_rules.h"

namespace lash {

namespace {

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t sgpr64(uint64_t v) { return ((uint64_t)sgpr((uint32_t)(v >> 32)) << 32) | sgpr((uint32_t)v); }
// a load the compiler cannot turn into a scalar one: SMEM returns out of order, so every LDS wait would also wait for a
// descriptor prefetch that is meant to be in flight for a whole genome
__device__ __forceinline__ uint64_t vload64(const uint64_t *p, uint64_t i)
{
    asm volatile("" : "+v"(i));
    return p[i];
}
__device__ __forceinline__ void store16_any(uint8_t *p, const uint4 v) { __builtin_memcpy(p, &v, 16); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 lds_load4(uint32_t byte_addr)
{
    const u32x4 r = *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)byte_addr;     // ds_read_b128
    return make_uint4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void lds_store4(uint32_t byte_addr, uint32_t v)
{
    const u32x4 r = {v, v, v, v};
    *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)byte_addr = r;                 // ds_write_b128
}

// the workgroup's barrier: a one-wave workgroup needs none (its LDS operations execute in order)
__device__ __forceinline__ void wg_barrier(uint32_t nw)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nw > 1u) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- the image of one genome out of the table, the table re-armed ---------------------------------------------------------------
// 16 bytes of image per lane and store: 8 HyperMinHash registers (u16) or 16 HyperLogLog / UltraLogLog registers (u8)
template <int ALGO>
__device__ __forceinline__ void sole_flush(const SoleArgs &a, uint32_t g, uint32_t hist_b, int p, uint32_t T)
{
    const uint32_t tid = threadIdx.x;
    uint8_t *img = a.images + (uint64_t)g * a.image_bytes;
    const uint32_t HDR = a.lay.hdr_bytes;
    uint8_t *regs_out = img + HDR;
    uint32_t *hist = (uint32_t *)(__attribute__((address_space(3))) uint32_t *)(uintptr_t)hist_b;
    HllTally tally;
    // HyperMinHash registers in image byte order (layout.hmh_reg_be): ONE byte permute per word with a selector from the layout — the
    // shift / shift / permute / select form cost 16 of the flush loop's 69 instructions for a switch that is off by default
    const uint32_t order_sel = (ALGO == 0 && a.lay.hmh_reg_be) ? 0x02030001u : 0x03020100u;
    auto img_order = [&](uint32_t v) { return ALGO == 0 ? __builtin_amdgcn_perm(v, v, order_sel) : v; };
    auto emit = [&](uint32_t i16, uint4 v) {                 // 16 image bytes at regs_out + 16 * i16
        uint8_t *dst = regs_out + 16ull * i16;
        if (a.accumulate) {
            const uint4 old = load16_any(dst);
            v.x = merge_word<ALGO>(img_order(old.x), v.x); v.y = merge_word<ALGO>(img_order(old.y), v.y);
            v.z = merge_word<ALGO>(img_order(old.z), v.z); v.w = merge_word<ALGO>(img_order(old.w), v.w);
        }
        if constexpr (ALGO == 1) { tally.add(hist, v.x); tally.add(hist, v.y); tally.add(hist, v.z); tally.add(hist, v.w); }
        if constexpr (ALGO == 0) { v.x = img_order(v.x); v.y = img_order(v.y); v.z = img_order(v.z); v.w = img_order(v.w); }
        store16_any(dst, v);
    };
    if constexpr (ALGO == 0) {
        // table word = (lz - 1) << 10 | sig under a signed maximum, -1 = empty  ->  register lz << 10 | sig, 0.  Two registers at a time in packed
        // 16-bit arithmetic (round 6: a 10 kbp genome spends a quarter of its instructions in this loop): the words' low halves side by side
        // (empty = 0xFFFF, real ones <= 50 << 10 | 0x3FF), t = w + 1 (empty -> 0), then t + 0x3FF * min(t, 1) = w + 0x400, or 0
        auto reg2 = [](uint32_t lo, uint32_t hi) {
            uint32_t w = __builtin_amdgcn_perm(hi, lo, 0x05040100u), t, u, r;
            const uint32_t one = 0x00010001u, k3ff = 0x03FF03FFu;
            asm("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(w), "s"(one));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(u) : "v"(t), "s"(one));
            asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(u), "s"(k3ff), "v"(t));
            return r;
        };
        for (uint32_t i = tid; i < HMH_M / 8u; i += T) {
            const uint4 t0 = lds_load4(32u * i), t1 = lds_load4(32u * i + 16u);
            lds_store4(32u * i, RANK_EMPTY); lds_store4(32u * i + 16u, RANK_EMPTY);
            emit(i, make_uint4(reg2(t0.x, t0.y), reg2(t0.z, t0.w), reg2(t1.x, t1.y), reg2(t1.z, t1.w)));
        }
        if (HDR && tid == 0) write_header(img, a.lay.hdr_tpl, a.alpha_bits, HMH_M, 0, 0.0, HMH_P);
    } else if constexpr (ALGO == 1) {
        // table word = rho - 1, -1 = empty  ->  register rho, 0
        auto reg4 = [](const uint4 t) { return ((t.x + 1u) & 0xFFu) | (((t.y + 1u) & 0xFFu) << 8) | (((t.z + 1u) & 0xFFu) << 16) | ((t.w + 1u) << 24); };
        const uint32_t n16 = (1u << p) >> 4;                  // p >= 4
        for (uint32_t i = tid; i < n16; i += T) {
            uint4 o;
            const uint32_t b = 64u * i;
            o.x = reg4(lds_load4(b)); o.y = reg4(lds_load4(b + 16u)); o.z = reg4(lds_load4(b + 32u)); o.w = reg4(lds_load4(b + 48u));
            lds_store4(b, RANK_EMPTY); lds_store4(b + 16u, RANK_EMPTY); lds_store4(b + 32u, RANK_EMPTY); lds_store4(b + 48u, RANK_EMPTY);
            emit(i, o);
        }
        tally.flush(hist);
        wg_barrier(T >> 6);
        if (tid < 64u) {
            if (tid == 0 && a.hll_corner) a.hll_corner[g] = 0u;            // (set again below if a register lies above 53 - p: same lane)
            write_hll_header_wave(img, a.lay.hdr_tpl, hist, a.alpha_bits, p, a.hll_corner ? a.hll_corner + g : nullptr);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            hist[tid] = 0u;                                   // (for the next genome)
            if (tid < 16u) hist[64u + tid] = 0u;
        }
        // ... whose tallies may come at once: two flushes follow each other without a hash pass between them when a genome has no k-mer
        // at all (an empty file, the empty half of an accumulating call) — the other waves' atomics then raced the first wave's zeros and
        // the second header's sum lost counts (registers right, 3 bytes of `sum` off; found by the randomized runner once LASH_SOLE_WGS
        // put several genomes on one workgroup)
        wg_barrier(T >> 6);
    } else {
        // table = 64-bit bitmaps of the nlz values seen -> hash4j's prefix (<< p - 1) -> pack(): 4 * (index of the leading one) + the
        // two bits below it
        auto reg1 = [&](uint32_t lo, uint32_t hi) -> uint32_t {
            if ((lo | hi) == 0u) return 0u;
            const uint64_t x = (((uint64_t)hi << 32) | lo) << (p - 1);
            const uint32_t top = 63u - (uint32_t)__builtin_clzll(x);
            const uint32_t below = top >= 2 ? (uint32_t)(x >> (top - 2)) & 3u : (uint32_t)(x << (2 - top)) & 3u;
            return (top << 2) | below;
        };
        auto reg4 = [&](uint32_t b) {
            const uint4 t0 = lds_load4(b), t1 = lds_load4(b + 16u);
            lds_store4(b, 0u); lds_store4(b + 16u, 0u);
            return reg1(t0.x, t0.y) | (reg1(t0.z, t0.w) << 8) | (reg1(t1.x, t1.y) << 16) | (reg1(t1.z, t1.w) << 24);
        };
        const uint32_t n16 = (1u << p) >> 4;
        for (uint32_t i = tid; i < n16; i += T) {
            const uint32_t b = 128u * i;
            uint4 o;
            o.x = reg4(b); o.y = reg4(b + 32u); o.z = reg4(b + 64u); o.w = reg4(b + 96u);
            emit(i, o);
        }
        if (p == 3 && tid < 2u) {                            // eight registers: two words
            uint8_t *dst = regs_out + 4u * tid;
            uint32_t v = reg4(32u * tid);
            if (a.accumulate) v = merge_word<2>(load_u32_any(dst), v);
            store_u32_any(dst, v);
        }
        if (tid == 0) write_header(img, a.lay.hdr_tpl, a.alpha_bits, 1ull << p, 0, 0.0, p);
    }
}

}  // namespace

template <int ALGO, int KMODE, bool XLOW, bool PACKED>
__global__ void __launch_bounds__(512) sole_sketch_kernel(SoleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_regs[];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_regs != 0u) __builtin_trap();   // the table starts at LDS address 0
    const uint32_t T = blockDim.x, tid = threadIdx.x, lane = tid & 63u, nw = T >> 6;
    const uint32_t wave = sgpr(tid >> 6);
    LdsRegs regs;
    regs.base = lds_regs;
    const int k = a.k, p = a.p;
    KParams kp;
    kp.bitflip = BitFlip::vector(a.bitflip);
    kp.p = p;
    kp.sh_lt = 32u - 2u * (uint32_t)k;
    kp.mask_lt = (KMODE == KM_LT16) ? ((1u << (2 * k)) - 1u) : 0xFFFFFFFFu;
    kp.sh_gt = 64u - 2u * (uint32_t)k;
    kp.mask_gt = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    kp.mask_hi = (uint32_t)(kp.mask_gt >> 32);
    kp.to_vector_registers();
    const uint32_t cmask = a.lay.comp_mask;
    const CodeTabs ct{a.lay.code_lo, a.lay.code_hi};
    const uint32_t ring_b = a.ring_off, brk_b = a.brk_off, scan_b = a.scan_off;
    const uint32_t RM = a.ring_words - 1u, BM = (a.ring_words >> 1) - 1u;
    constexpr bool K21 = KMODE == KM_GT16;                                 // a k = 21 body of its own (canon_gt16<21>)

    // everything clear, once: the table armed, rings and histogram zero (each genome leaves them that way)
    for (uint32_t i = tid; i < a.lds_words; i += T) lds_regs[i] = (i < a.nreg32 && ALGO != 2) ? RANK_EMPTY : 0u;
    wg_barrier(nw);

    unsigned long long kmers_wave = 0, bases_wg = 0;                        // valid k-mers this WAVE hashed; surviving bases (the same in every wave)
    uint32_t parity = 0;

    // ---- one pass over ring words [hw, hw + n) (n <= T): lane tid hashes word hw + tid; nk = k-mer starts of the whole genome when the
    //      genome is complete (masks its end), else ~0 ----
    uint32_t zero_w = 0, zero_n = 0, bzero = 0;                            // ring words the previous pass consumed (zeroed by the next one); first break word not yet zeroed
    auto hash_pass = [&](uint32_t hw, uint32_t n, uint32_t nk, bool breaks, bool fine) {
        // words consumed by the previous pass are free again (appends OR into zeroed words); a barrier lies between the two passes.
        // A break word covers TWO code words: it is free once both have been hashed
        if (tid < zero_n) lds_store(ring_b + 4u * ((zero_w + tid) & RM), 0u);
        {
            const uint32_t bz_end = (zero_w + zero_n) >> 1;
            if (breaks && bzero + tid < bz_end) lds_store(brk_b + 4u * ((bzero + tid) & BM), 0u);
            bzero = bz_end > bzero ? bz_end : bzero;
        }
        zero_w = hw; zero_n = n;
        // `fine` (the genome's last words): few words are handed out a QUARTER (4 k-mer starts) or a half per lane, so that what is left
        // after the full passes costs a chain of 4 or 8 k-mers, not 16 — and a 1 kbp genome spreads over 250 lanes instead of 62
        const uint32_t upl = !fine || 2u * n > T ? 4u : (4u * n > T ? 2u : 1u);                  // quarter units per lane
        const uint32_t u0 = upl * tid;                                                            // the lane's first unit of the pass
        const bool active = u0 < 4u * n;
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) return;
        const uint32_t w = hw + (u0 >> 2), pos0 = 16u * w;
        const uint32_t junk = (tid + 1u) * 0x9E3779B1u;
        uint32_t c0 = junk, c1 = ~junk, c2 = junk;
        uint32_t kvw = 0;
        if (active) {
            c0 = lds_load(ring_b + 4u * (w & RM)); c1 = lds_load(ring_b + 4u * ((w + 1u) & RM));
            if constexpr (KMODE == KM_GT16) c2 = lds_load(ring_b + 4u * ((w + 2u) & RM));
            kvw = 0xFFFFu;                                                  // a complete word of a genome without record starts: 16 k-mers
            if (breaks || nk != 0xFFFFFFFFu) {
                uint32_t b0 = 0, b1 = 0;
                if (breaks) {
                    const uint32_t bw = w >> 1;
                    const uint32_t x0 = lds_load(brk_b + 4u * (bw & BM)), x1 = lds_load(brk_b + 4u * ((bw + 1u) & BM)), x2 = lds_load(brk_b + 4u * ((bw + 2u) & BM));
                    b0 = (w & 1u) ? (x0 >> 16) | (x1 << 16) : x0;
                    b1 = (w & 1u) ? (x1 >> 16) | (x2 << 16) : x1;
                }
                kvw = (uint32_t)kmer_valid_mask(b0, b1, 0u, pos0, nk, k) & 0xFFFFu;
            }
        }
        const uint32_t r0 = rcword(c0, cmask), r1 = rcword(c1, cmask), r2 = (KMODE == KM_GT16) ? rcword(c2, cmask) : 0u;
        if (upl < 4u) {
            // ---- quarters ----
            const uint32_t rq = 4u * (u0 & 3u);
            kvw &= ((1u << (4u * upl)) - 1u) << rq;                          // the lane's own positions
            kmers_wave += wave_sum((uint32_t)__builtin_popcount(kvw));
            for (uint32_t i = 0; i < upl; ++i) {
                uint32_t m = kvw;
                asm volatile("" : "+v"(m));
                const uint32_t z = process_quarter<ALGO, KMODE, XLOW, true>(regs, kp, c0, c1, c2, r0, r1, r2, m, rq + 4u * i);
                constexpr uint32_t Z_REDO = z_redo<ALGO, LdsRegs>();
                if (z <= Z_REDO) (void)process_quarter<ALGO, KMODE, XLOW, false>(regs, kp, c0, c1, c2, r0, r1, r2, m, rq + 4u * i);
            }
            return;
        }
        const bool all_valid = __builtin_amdgcn_ballot_w64(kvw != 0xFFFFu) == 0ull;
        kmers_wave += all_valid ? 1024u : wave_sum((uint32_t)__builtin_popcount(kvw));
        uint32_t z;
        if (K21 && k == 21) {
            if (all_valid) z = process_word<ALGO, KMODE, XLOW, false, true, LdsRegs, K21 ? 21 : 0>(regs, kp, c0, c1, c2, r0, r1, r2, 0u);
            else {
                uint32_t m = kvw;
                asm volatile("" : "+v"(m));
                z = process_word<ALGO, KMODE, XLOW, true, true, LdsRegs, K21 ? 21 : 0>(regs, kp, c0, c1, c2, r0, r1, r2, m);
            }
        } else if (all_valid) {
            z = process_word<ALGO, KMODE, XLOW, false, true>(regs, kp, c0, c1, c2, r0, r1, r2, 0u);
        } else {
            uint32_t m = kvw;
            asm volatile("" : "+v"(m));
            z = process_word<ALGO, KMODE, XLOW, true, true>(regs, kp, c0, c1, c2, r0, r1, r2, m);
        }
        // the fast forms return a word whose smallness says "rank not decided by the bits looked at": the exact form again (idempotent)
        constexpr uint32_t Z_REDO = z_redo<ALGO, LdsRegs>();
        if (z <= Z_REDO) {
            uint32_t m = kvw;
            asm volatile("" : "+v"(m));
            (void)process_word<ALGO, KMODE, XLOW, true, false>(regs, kp, c0, c1, c2, r0, r1, r2, m);
        }
    };

    // ---- chunks of genomes ----
    uint32_t chunk = blockIdx.x;
    while (chunk < a.n_chunks) {
        // the next chunk, asked for now (one returning atomic per chunk; its round trip runs under this chunk's work)
        uint32_t next_chunk = 0;
        if (tid == 0) next_chunk = gridDim.x + atomicAdd(a.ticket, 1u);
        const uint32_t g_begin = a.chunk_begin[chunk], g_end = a.chunk_begin[chunk + 1];
        if constexpr (!PACKED) {
            // genome g = bytes [b0, b1) of seq; the offsets of g + 2 are asked for when g begins
            uint32_t g = g_begin;
            uint64_t b0 = sgpr64(vload64(a.genome_byte_off, g)), b1 = sgpr64(vload64(a.genome_byte_off, g + 1u));
            uint64_t p_b = vload64(a.genome_byte_off, g + 2u <= g_end ? g + 2u : g_end);
            // The round in flight: 16 bytes per lane from absolute byte q_at + 16 * tid, and their record-start bits.  Addresses are a
            // uniform base + a 32-bit lane offset, clamped so that the 16 bytes stay inside the buffer (seq_bytes >= 16): only the lanes
            // of the buffer's very last round read somewhere else than they meant to, and shift their bytes into place (q_fix)
            uint64_t q_at = ~0ull;
            uint4 q_nxt = make_uint4(0, 0, 0, 0);
            uint32_t rb_nxt = 0;
            const uint32_t lane_off = 16u * tid;
            const uint64_t last16 = a.seq_bytes - 16;
            auto round_load = [&](uint64_t at, uint4 &q, uint32_t &rbw) {
                if (__builtin_expect(at + 16ull * T <= a.seq_bytes, 1)) {
                    q = load16_any(a.seq + at + lane_off);                                   // uniform base + lane offset
                } else {
                    const uint64_t base = at < last16 ? at : last16;                         // uniform
                    const uint32_t d = (uint32_t)(at - base), cap = last16 - base > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(last16 - base);
                    uint32_t vo = lane_off + d;
                    vo = vo < cap ? vo : cap;
                    q = load16_any(a.seq + base + vo);
                }
                rbw = 0;
                if (a.brk_abs) {
                    // 4 bytes of the bitmap at any alignment: bits 8 * (A >> 3) .. + 31 hold the lane's 16 (the bitmap is padded by a round)
                    uint32_t x;
                    __builtin_memcpy(&x, reinterpret_cast<const uint8_t *>(a.brk_abs) + (at >> 3) + 2u * tid, 4);
                    rbw = x;
                }
            };
            // the lane meant to read 16 bytes from `at + lane_off` and read them `shift` bytes earlier: bring them down, 'N' behind them
            auto q_fix = [&](uint64_t at, uint4 &q) {
                const uint64_t base = at < last16 ? at : last16;
                const uint32_t d = (uint32_t)(at - base), cap = last16 - base > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(last16 - base);
                if (cap >= lane_off + d) return;                                              // (this lane read what it meant to)
                const uint32_t shift = lane_off + d - cap;                                   // bytes
                uint64_t lo = (uint64_t)q.x | ((uint64_t)q.y << 32), hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
                const uint64_t fill = 0x4E4E4E4E4E4E4E4Eull;
                if (shift >= 16u) { lo = fill; hi = fill; }
                else if (shift >= 8u) { const uint32_t s8 = 8u * (shift - 8u); lo = s8 ? (hi >> s8) | (fill << (64u - s8)) : hi; hi = fill; }
                else { const uint32_t s8 = 8u * shift; lo = (lo >> s8) | (hi << (64u - s8)); hi = (hi >> s8) | (fill << (64u - s8)); }
                q = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
            };
            const bool breaks = a.brk_abs != nullptr;
            // The genome whose registers are complete but still in the table.  Its image is written when the NEXT genome is about to hash
            // its first word, not when its own last word is done: a wave's loads and stores share one counter (vmcnt), so a wave that has
            // just issued the image's stores and then waits for the next genome's bytes waits for the stores to reach memory — 2 us per
            // genome with nothing to do.  Flushed late, the stores drain under the next genome's first hash pass.
            constexpr uint32_t NONE = 0xFFFFFFFFu;
            uint32_t pending = NONE;
            for (; g < g_end; ++g) {
                const uint64_t Lb64 = b1 - b0;
                if (Lb64 <= a.max_len) {
                    const uint32_t Lb = (uint32_t)Lb64;
                    uint32_t N = 0, H = 0, rel = 0;                                         // bases appended; ring words hashed; bytes staged
                    bool whole = false;                                                     // the genome is in the ring to its last base
                    uint32_t limit = 0, nk = 0xFFFFFFFFu;                                   // ring words that may be hashed; k-mer starts (known at the end)
                    zero_w = 0; zero_n = 0; bzero = 0;                      // (zero_w too: a stale one made the first pass wipe this genome's early record starts)
#ifdef LASH_DEBUG_STALE
                    // Debug build (tools/build_debug_stale.sh): what a genome inherits must be EXACTLY the rest state — both
                    // rings zero from end to end, and the table armed unless the previous genome's registers still wait in it.  Anything else is
                    // a stale word that the release build would "mostly" survive (appends OR into it, a max against it): here it traps, and the
                    // call fails with a HIP error instead of an image that is right for most genomes.
                    wg_barrier(nw);
                    for (uint32_t i = tid; i < a.ring_words; i += T) if (lds_load(ring_b + 4u * i) != 0u) __builtin_trap();
                    for (uint32_t i = tid; i < (a.ring_words >> 1); i += T) if (lds_load(brk_b + 4u * i) != 0u) __builtin_trap();
                    if (pending == NONE)
                        for (uint32_t i = tid; i < a.nreg32; i += T) if (lds_regs[i] != (ALGO == 2 ? 0u : RANK_EMPTY)) __builtin_trap();
                    if (ALGO == 1 && tid < 72u && lds_load(a.hist_off + 4u * tid) != 0u && pending == NONE) __builtin_trap();
                    wg_barrier(nw);
#endif
                    for (;;) {
                        if (rel < Lb) {
                            // ---- stage one round: 16 bytes per lane -> survivors -> ring ----
                            const uint64_t at = b0 + rel;
                            if (q_at != at) { round_load(at, q_nxt, rb_nxt); q_at = at; }   // (first round of a chunk, or after a genome that was not ours)
                            uint4 q = q_nxt;
                            const uint32_t rbw = rb_nxt;
                            if (at + 16ull * T > a.seq_bytes) q_fix(at, q);                 // the buffer's last round
                            const bool inner = rel + 16u * T <= Lb;                         // every lane's 16 bytes are the genome's
                            // the next round's loads: this genome's next 16 bytes per lane, or the first bytes of the next genome
                            {
                                const uint64_t nat = inner && rel + 16u * T < Lb ? at + 16ull * T : b1;
                                round_load(nat, q_nxt, rb_nxt);
                                q_at = nat;
                            }
                            uint32_t bad = 0;
                            const uint32_t codes = ascii16_to_word(q, bad, ct);
                            // bytes of the lane's 16 that are the genome's: clamp(Lb - rel - 16 tid, 0, 16) of them (lengths are below 2^31)
                            uint32_t v;
                            {
                                int own = (int)(Lb - rel) - (int)lane_off;
                                asm("v_med3_i32 %0, %1, 0, 16" : "=v"(own) : "v"(own));
                                asm("v_bfm_b32 %0, %1, 0" : "=v"(v) : "v"(own));           // (1 << own) - 1
                            }
                            const uint32_t ownmask = v;
                            if (bad) v &= ~inv16(q);
                            const uint32_t rb = breaks ? (rbw >> ((uint32_t)at & 7u)) & ownmask : 0u;
                            // survivors before this lane's in the wave, in the workgroup
                            uint32_t n = 16u, off = 16u * lane, wtot = 1024u;
                            if (__builtin_amdgcn_ballot_w64(v != 0xFFFFu) != 0ull) {
                                n = (uint32_t)__builtin_popcount(v);
                                off = wave_excl_scan(n, wtot);
                            }
                            uint32_t base = 0, tot = wtot;
                            if (nw > 1u) {
                                if (lane == 0) lds_store(scan_b + 4u * (parity * 8u + wave), wtot);
                                wg_barrier(nw);
                                uint32_t t = lds_load(scan_b + 4u * (parity * 8u + (lane & 7u)));
                                t = lane < nw ? t : 0u;
                                uint32_t incl = t;                                          // prefix over lanes 0..7 (one DPP row)
                                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);
                                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);
                                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);
                                tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 7);
                                base = (uint32_t)__builtin_amdgcn_readlane((int)(incl - t), (int)wave);
                                parity ^= 1u;
                            }
                            // ---- append ----
                            const uint32_t P = N + base + off;
                            if (wtot == 1024u && ((N + base) & 15u) == 0u) {
                                lds_store(ring_b + 4u * ((P >> 4) & RM), codes);            // nothing deleted, word-aligned: the word as it is
                            } else if (n) {
                                uint32_t cb;
                                const uint32_t bits = compact16(codes, v, 0u, cb);
                                const uint32_t w = P >> 4, sh = 2u * (P & 15u);
                                lds_or(ring_b + 4u * (w & RM), bits >> sh);
                                if (sh && (P & 15u) + n > 16u) lds_or(ring_b + 4u * ((w + 1u) & RM), bits << (32u - sh));
                            }
                            if (rb) {
                                // a record start lands on the first surviving base at or after it: position = survivors before its byte
                                uint32_t m = rb;
                                do {
                                    const uint32_t j = (uint32_t)__builtin_ctz(m);
                                    m &= m - 1u;
                                    const uint32_t pos = P + (uint32_t)__builtin_popcount(v & ((1u << j) - 1u));
                                    lds_or(brk_b + 4u * ((pos >> 5) & BM), 1u << (pos & 31u));
                                } while (m);
                            }
                            // the words that were complete BEFORE this round may be hashed now: their last start's k - 1 followers were
                            // appended a barrier ago
                            const uint32_t need = (uint32_t)k + 15u;
                            limit = N >= need ? (N - need) / 16u + 1u : 0u;
                            N += tot;
                            rel += 16u * T;
                        } else if (!whole) {
                            // ---- the genome is in the ring: what is left, under its end's mask ----
                            wg_barrier(nw);
                            whole = true;
                            nk = N >= (uint32_t)k ? N - (uint32_t)k + 1u : 0u;
                            limit = (nk + 15u) >> 4;
                        }
                        if (H < limit) {
                            if (pending != NONE) {                                          // the table changes hands
                                sole_flush<ALGO>(a, pending, a.hist_off, p, T);
                                pending = NONE;
                                wg_barrier(nw);
                            }
                            const uint32_t cnt = limit - H < T ? limit - H : T;
                            hash_pass(H, cnt, nk, breaks, whole);
                            H += cnt;
                            if (whole) zero_n = 0;                                          // (no barrier between the last passes; the rings are wiped below)
                        }
                        if (whole && H >= limit) break;
                    }
                    bases_wg += N;
                    if (a.ndel && tid == 0) a.ndel[g] = Lb - N;
                    wg_barrier(nw);                                                         // every wave's k-mers are in the table, every wave is done with the rings
                    // the rings, zero again up to where this genome reached
                    if (Lb) {
                        const uint32_t top = (N >> 4) + 3u < a.ring_words ? (N >> 4) + 3u : a.ring_words;
                        for (uint32_t i = 4u * tid; i < top; i += 4u * T) lds_store4(ring_b + 4u * i, 0u);
                        if (breaks) for (uint32_t i = 4u * tid; i < (top >> 1) + 4u && i < (a.ring_words >> 1); i += 4u * T) lds_store4(brk_b + 4u * i, 0u);
                    }
                    // (a genome without a single k-mer never touched the table: the one before it is still owed its image)
                    if (pending != NONE) sole_flush<ALGO>(a, pending, a.hist_off, p, T);
                    pending = g;
                }
                // the next genome begins where this one ends
                b0 = b1;
                b1 = sgpr64(p_b);
                p_b = vload64(a.genome_byte_off, g + 3u <= g_end ? g + 3u : g_end);
            }
            if (pending != NONE) sole_flush<ALGO>(a, pending, a.hist_off, p, T);            // the chunk's last genome
        }
        // (PACKED: below)
        if constexpr (PACKED) {
            for (uint32_t g = g_begin; g < g_end; ++g) {
                const GenomeDesc gd = a.genomes[g];
                const uint64_t L64 = a.nvalid[g];
                if (gd.byte_len > a.max_len) continue;
                const uint32_t L = (uint32_t)L64;
                const bool breaks = gd.format != 0u || gd.rec_end - gd.rec_begin > 1;
                const uint32_t *__restrict__ w = a.words + gd.word_off;
                const uint32_t *__restrict__ bk = a.brk + gd.brk_off;
                const uint32_t n_words = (L + 15u) >> 4;
                uint32_t N = 0, H = 0;
                zero_w = 0; zero_n = 0; bzero = 0;                      // (zero_w too: a stale one made the first pass wipe this genome's early record starts)
                for (uint32_t w0 = 0; w0 < n_words; w0 += T) {
                    // the pack stage's stream is what the ring would hold: copy T words (and their break bits) in
                    const uint32_t wi = w0 + tid;
                    if (wi < n_words) {
                        uint32_t c = w[wi];
                        const uint32_t left = L - 16u * wi;
                        if (left < 16u) c &= ~(0xFFFFFFFFu >> (2u * left));                 // (bases past the genome's end: zero)
                        lds_store(ring_b + 4u * (wi & RM), c);
                        if (breaks && !(wi & 1u)) lds_store(brk_b + 4u * ((wi >> 1) & BM), bk[wi >> 1]);
                    }
                    const uint32_t N_prev = N;
                    N = 16u * (w0 + T) < L ? 16u * (w0 + T) : L;
                    wg_barrier(nw);
                    if (w0 + T < n_words) {
                        const uint32_t need = (uint32_t)k + 15u;
                        const uint32_t done = N_prev >= need ? (N_prev - need) / 16u + 1u : 0u;
                        if (done > H) {
                            const uint32_t cnt = done - H < T ? done - H : T;
                            hash_pass(H, cnt, 0xFFFFFFFFu, breaks, false);
                            H += cnt;
                        }
                    }
                }
                // (the last copy's barrier stands; plain stores need no zeroed words, but hash_pass's zeroing must not hit live ones)
                const uint32_t nk = N >= (uint32_t)k ? N - (uint32_t)k + 1u : 0u, nk_words = (nk + 15u) >> 4;
                zero_n = 0;
                while (H < nk_words) {
                    const uint32_t cnt = nk_words - H < T ? nk_words - H : T;
                    hash_pass(H, cnt, nk, breaks, true);
                    H += cnt;
                    zero_n = 0;
                }
                bases_wg += N;
                wg_barrier(nw);
                sole_flush<ALGO>(a, g, a.hist_off, p, T);
                // (no wipe: every ring word and break word a VALID k-mer of the next genome reads is copied in by that genome first;
                //  its stores cannot overtake this genome's reads, which lie before the barrier above)
            }
        }
        // every wave learns the next chunk (thread 0's atomic has had the whole chunk to return)
        if (tid == 0) lds_store(scan_b + 4u * 16u, next_chunk);
        wg_barrier(nw);
        chunk = sgpr(lds_load(scan_b + 4u * 16u));
        wg_barrier(nw);
    }

    // ---- this workgroup's census: k-mers (summed over its waves), surviving bases ----
    {
        unsigned long long *slot = a.wg_counts + 2ull * blockIdx.x;
        if (lane == 0) {
            lds_store(scan_b + 4u * (2u * wave), (uint32_t)kmers_wave);
            lds_store(scan_b + 4u * (2u * wave + 1u), (uint32_t)(kmers_wave >> 32));
        }
        wg_barrier(nw);
        if (tid == 0) {
            unsigned long long tot = 0;
            for (uint32_t w = 0; w < nw; ++w) tot += (unsigned long long)lds_load(scan_b + 8u * w) | ((unsigned long long)lds_load(scan_b + 8u * w + 4u) << 32);
            slot[0] = tot;
            slot[1] = bases_wg;
        }
    }
}

// ---- record starts as bits at absolute byte positions -------------------------------------------------------------------------
// A record's first base is a barrier for k-mers (utils.rs:457-464).  Every record start is marked, a genome's first record included:
// no k-mer of the genome begins before it, so its bit (position 0 of the ring, or the first survivor's) masks nothing.
__global__ void __launch_bounds__(256) sole_mark_kernel(const uint64_t *rec_off, uint64_t n_rec, uint64_t seq_bytes, uint32_t *brk_abs)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = rec_off[r];
        if (b < seq_bytes) atomicOr(brk_abs + (b >> 5), 1u << (b & 31u));
    }
}

__global__ void __launch_bounds__(256) sole_census_kernel(const unsigned long long *wg_counts, uint32_t n_wg, unsigned long long *counter,
                                                          unsigned long long *bases, uint32_t *ticket)
{
    if (threadIdx.x == 0) *ticket = 0u;                                     // at rest the chunk ticket is zero: no memset per launch
    // one workgroup: the k-mer census is ADDED to the context's running count, the surviving bases are those of this call
    __shared__ unsigned long long part[2][4];
    unsigned long long km = 0, bs = 0;
    for (uint32_t i = threadIdx.x; i < n_wg; i += blockDim.x) { km += wg_counts[2ull * i]; bs += wg_counts[2ull * i + 1]; }
    for (int off = 32; off > 0; off >>= 1) { km += __shfl_down(km, off, 64); bs += __shfl_down(bs, off, 64); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = km; part[1][threadIdx.x >> 6] = bs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        km = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        if (km) atomicAdd(counter, km);
        *bases = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}

// ------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------------------
SolePlan make_sole_plan(int algo, int p, uint32_t n_genomes, uint32_t cu_count)
{
    SolePlan s{};
    uint32_t table;
    if (algo == 0) table = HMH_M * 4u;
    else if (algo == 1) table = 4u << p;
    else table = 8u << p;
    s.ok = table <= 64u * 1024u && !(algo == 1 && (p < 4 || p > 16)) && !(algo == 2 && (p < 3 || p > 26));
    if (!s.ok) return s;
    // as many waves per CU as the registers allow (4 per SIMD = 16 per CU) in as FEW waves per workgroup as the table's LDS admits:
    // a small table (HyperLogLog p = 10: 4 KiB) makes every wave a workgroup with a genome of its own, and barriers no-ops
    auto layout = [&](uint32_t threads, SolePlan &o) {
        o.threads = threads;
        o.hist_off = table;
        o.scan_off = o.hist_off + 80u * 4u;
        o.ring_words = 4u * threads;                                        // four rounds' worth of bases
        o.ring_off = (o.scan_off + 20u * 4u + 15u) & ~15u;
        o.brk_off = o.ring_off + o.ring_words * 4u;
        o.lds_bytes = o.brk_off + (o.ring_words / 2u) * 4u + 16u;
        // (16 waves per CU: the kernels take 64 .. 128 vector registers; sole_resident_per_cu() asks the runtime for the launch itself)
        o.wg_per_cu = std::min(std::min(32u, (160u * 1024u) / o.lds_bytes), 1024u / threads);
    };
    // ... as long as every workgroup still gets a few genomes: 2 000 genomes on 4 096 one-wave workgroups would leave half the
    // chip idle (and each genome to a single wave)
    uint32_t best = 512;
    for (uint32_t t : {64u, 128u, 256u, 512u}) {
        SolePlan o{};
        layout(t, o);
        if (o.wg_per_cu * (t / 64u) < 16u) continue;
        if (n_genomes && (uint64_t)o.wg_per_cu * cu_count * 4u > n_genomes && t < 512u) continue;
        best = t;
        break;
    }
    if (const char *e = getenv("LASH_SOLE_THREADS")) {                      // tuning knob (tools/)
        const int t = atoi(e);
        if (t == 64 || t == 128 || t == 256 || t == 512) best = (uint32_t)t;
    }
    layout(best, s);
    s.ok = true;
    return s;
}

// n_wg == 0: no launch — *occ receives the workgroups of this variant that are resident on one CU at a time (registers, LDS and wave
// slots taken together: the persistent launch must not be larger than what is resident, its chunks are handed out to running workgroups)
template <int ALGO, int KMODE, bool XLOW, bool PACKED>
static hipError_t launch_sole_one(const SolePlan &plan, const SoleArgs &args, uint32_t n_wg, hipStream_t stream, uint32_t *occ)
{
    auto kern = sole_sketch_kernel<ALGO, KMODE, XLOW, PACKED>;
    if (plan.lds_bytes > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (occ) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(kern), (int)plan.threads, plan.lds_bytes);
        if (e != hipSuccess) return e;
        *occ = (uint32_t)std::max(1, n);
        return hipSuccess;
    }
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(plan.threads), plan.lds_bytes, stream, args);
    return hipGetLastError();
}
template <int ALGO, bool XLOW, bool PACKED>
static hipError_t launch_sole_kmode(const SolePlan &plan, int k, const SoleArgs &args, uint32_t n_wg, hipStream_t stream, uint32_t *occ)
{
    if (k == 16) return launch_sole_one<ALGO, KM_16, XLOW, PACKED>(plan, args, n_wg, stream, occ);
    if (k < 16) return launch_sole_one<ALGO, KM_LT16, XLOW, PACKED>(plan, args, n_wg, stream, occ);
    return launch_sole_one<ALGO, KM_GT16, XLOW, PACKED>(plan, args, n_wg, stream, occ);
}
template <bool PACKED>
static hipError_t launch_sole_algo(const SolePlan &plan, int algo, int k, bool x_low, const SoleArgs &args, uint32_t n_wg, hipStream_t stream, uint32_t *occ)
{
    switch (algo) {
    case 0: return x_low ? launch_sole_kmode<0, true, PACKED>(plan, k, args, n_wg, stream, occ) : launch_sole_kmode<0, false, PACKED>(plan, k, args, n_wg, stream, occ);
    case 1: return x_low ? launch_sole_kmode<1, true, PACKED>(plan, k, args, n_wg, stream, occ) : launch_sole_kmode<1, false, PACKED>(plan, k, args, n_wg, stream, occ);
    case 2: return launch_sole_kmode<2, false, PACKED>(plan, k, args, n_wg, stream, occ);
    default: return hipErrorInvalidValue;
    }
}
hipError_t launch_sole(const SolePlan &plan, int algo, int k, bool x_low, bool packed, const SoleArgs &args, uint32_t n_wg, hipStream_t stream)
{
    if (n_wg == 0 || !plan.ok) return n_wg ? hipErrorInvalidValue : hipSuccess;
    SoleArgs a = args;
    a.hist_off = plan.hist_off; a.scan_off = plan.scan_off; a.ring_off = plan.ring_off; a.brk_off = plan.brk_off;
    a.ring_words = plan.ring_words; a.lds_words = plan.lds_bytes / 4u;
    return packed ? launch_sole_algo<true>(plan, algo, k, x_low, a, n_wg, stream, nullptr) : launch_sole_algo<false>(plan, algo, k, x_low, a, n_wg, stream, nullptr);
}

hipError_t sole_resident_per_cu(const SolePlan &plan, int algo, int k, bool x_low, bool packed, uint32_t *out)
{
    if (!plan.ok) return hipErrorInvalidValue;
    // asked once per kernel variant and workgroup shape
    // (one entry = {occupancy, LDS bytes it was asked with} in ONE atomic word: `lash sketch --gpus N` runs a thread per device through here)
    static std::atomic<uint64_t> cache[2][3][3][2][17] = {};
    if (algo < 0 || algo > 2 || plan.threads / 64u > 16u) return hipErrorInvalidValue;
    std::atomic<uint64_t> &c = cache[packed ? 1 : 0][algo][k == 16 ? 0 : k < 16 ? 1 : 2][x_low ? 1 : 0][plan.threads / 64u];
    uint64_t e64 = c.load(std::memory_order_acquire);
    if ((uint32_t)e64 == 0u || (uint32_t)(e64 >> 32) != plan.lds_bytes) {
        SoleArgs none{};
        uint32_t occ = 0;
        hipError_t e = packed ? launch_sole_algo<true>(plan, algo, k, x_low, none, 0, nullptr, &occ) : launch_sole_algo<false>(plan, algo, k, x_low, none, 0, nullptr, &occ);
        if (e != hipSuccess) return e;
        e64 = ((uint64_t)plan.lds_bytes << 32) | occ;
        c.store(e64, std::memory_order_release);
    }
    *out = (uint32_t)e64;
    return hipSuccess;
}

hipError_t launch_sole_mark(const uint64_t *rec_off, uint64_t n_rec, uint64_t seq_bytes, uint32_t *brk_abs, hipStream_t stream)
{
    if (n_rec == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(4096, (n_rec + 255) / 256);
    hipLaunchKernelGGL(sole_mark_kernel, dim3(blocks), dim3(256), 0, stream, rec_off, n_rec, seq_bytes, brk_abs);
    return hipGetLastError();
}

hipError_t launch_sole_census(const unsigned long long *wg_counts, uint32_t n_wg, unsigned long long *counter, unsigned long long *bases,
                              uint32_t *ticket, hipStream_t stream)
{
    hipLaunchKernelGGL(sole_census_kernel, dim3(1), dim3(256), 0, stream, wg_counts, n_wg, counter, bases, ticket);
    return hipGetLastError();
}

}  // namespace lash
