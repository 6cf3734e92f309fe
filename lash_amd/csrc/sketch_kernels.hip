// sketch_kernels.hip — the dominant kernel of liblash_gfx950: canonical k-mers -> xxh3 -> register update.
//
// Replaces the three `while let Some(km) = it.next()` loops of the reference
// (/root/reference/src/utils.rs:469-476, 481-488, 493-498) together with the add_kmer bodies they call
// (utils.rs:395-398 HMH, 411-413 HLL, 427-429 ULL).  Written for gfx950 only (wave64, LDS atomics, v_alignbit).
//
// Shape of the computation (DESIGN.md "Kernels"):
//   * one workgroup owns one slice of one genome and a private copy of the sketch in LDS
//     (HMH 16384 x u32 = 64 KiB -> two workgroups per CU; HLL 2^p x u32; ULL 2^p x u64 bitmaps);
//   * a lane owns 4 consecutive packed words (64 k-mer start positions) per step, so one wave reads 1 KiB of
//     contiguous 2-bit bases with one global_load_dwordx4 per lane; look-ahead words come from the same lines;
//   * a k-mer is a funnel-shift window: fwd = v_alignbit(w[j], w[j+1], 32-2r); its reverse complement is the
//     mirrored window of the per-word reverse-complemented stream: rc = v_alignbit(RC[j+1], RC[j], 2r).
//     Nothing rolls, so the 16 positions of a word are independent instruction streams (ILP for the multiplies);
//   * the register update is a fire-and-forget LDS atomic (ds_max_u32 / ds_or_b32).  max and OR are commutative
//     and idempotent, so lanes, waves, workgroups and GPUs may take any partition of the k-mer multiset
//     (SURVEY.md §7.3); partial sketches of a genome's slices are reduced by finalize_kernel.
//   * k-mers that would span two records, or run past the genome's end, are neutralised by AND-ing the update
//     value with a 0/~0 mask taken from the record-break bitmap: max(x,0) and OR 0 are no-ops (no divergence).
#include "sketch_rules.h"

namespace lash {

// XLOW: the rule's variant (HyperMinHash x = low half; HyperLogLog bucket = top bits; sketch_rules.h, add_kmer)
template <int ALGO, int KMODE, bool XLOW, int REGS, bool DIRECT, bool DEFER = false>
__global__ void __launch_bounds__(1024) LASH_SKETCH_WAVES_PER_EU_ATTR sketch_kernel(SketchArgs a)
{
    static_assert(!DEFER || (ALGO == 0 && REGS == REGS_LDS), "deferred signatures: HyperMinHash, one LDS table");
    // dynamic LDS: [nreg32 register words][16 words of per-wave census]; registers start at LDS offset 0 so the
    // bucket offset goes straight into the ds_max / ds_or address
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_regs[];

    // work items are handed out longest first when their sizes differ (a.item_order: the host's permutation), so that the last
    // workgroups to start are the short ones: a mixed collection lost 11 % to its tail in launch order = genome order
    const uint32_t slot = xcd_skewed(blockIdx.x, gridDim.x);
    const uint32_t item = a.item_order ? a.item_order[slot] : slot + a.item_base;
    const WorkItem it = a.items[item];
#ifdef LASH_ITEM_TRACE_BUILD   // diagnostic build only (tools/build_trace_lib.sh): in the shipped kernels the two branches cost scalar registers the tile loop is short of
    if (a.item_trace && threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.item_trace[4ull * item] = wall_clock64();
        a.item_trace[4ull * item + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const GenomeDesc gd = a.genomes[it.genome];
    const uint64_t L = DIRECT ? gd.byte_len : a.nvalid[it.genome];
    const int k = a.k, p = a.p;
    const uint64_t nk = L >= (uint64_t)k ? L - (uint64_t)k + 1 : 0;     // k-mer start positions of the genome
    // (a genome is shorter than 2^32 - 64 bytes: the per-tile tests are done on 32-bit word counts — one scalar register each
    // instead of a pair and a 64-bit compare; the loop is short of scalar registers)
    const uint32_t nk_words = (uint32_t)((nk + 15) >> 4);                // word w holds a k-mer start <=> w < nk_words
    const uint32_t fast_words = L >= 96 ? (uint32_t)((L - 96) >> 4) + 1u : 0u;   // all 96 bytes from word w on are inside <=> w < fast_words
    if ((uint64_t)it.word_begin * 16 >= nk) {                             // slice beyond the surviving bases
        // the only work item of a genome too short for a single k-mer still owes the (empty) image; when the call
        // unions into existing images there is nothing to add
        if ((it.slice & ITEM_SOLE) && !a.accumulate) {
            uint8_t *img = a.images + (uint64_t)it.genome * a.image_bytes;
            for (uint64_t i = threadIdx.x; i < a.image_bytes; i += blockDim.x) img[i] = 0;
            __syncthreads();
            if (threadIdx.x == 0) {                                          // all registers 0: zero = m, sum = m * 2^-0
                const uint64_t n_regs = ALGO == 0 ? HMH_M : (1ull << p);
                write_header(img, a.lay.hdr_tpl, a.alpha_bits, n_regs, n_regs, (double)n_regs, ALGO == 0 ? HMH_P : p);
            }
        }
        return;
    }

    constexpr bool USE_LDS = REGS != REGS_GLOBAL;
    using Regs = typename std::conditional<DEFER, typename std::conditional<XLOW, LdsThrRegsX, LdsThrRegs>::type,
                                           typename std::conditional<REGS == REGS_LDS, LdsRegs,
                                           typename std::conditional<REGS == REGS_GLOBAL, GlobalRegs,
                                           typename std::conditional<REGS == REGS_BINS, BinRegs, LdsByteQRegs<ALGO>>::type>::type>::type>::type;
    // what the slow paths (junction walks, dense tiles) update through: the byte tables' plain compare-and-swap form
    using SlowRegs = typename std::conditional<REGS == REGS_BYTES, LdsByteRegs, Regs>::type;
    Regs regs;
    uint32_t *census;
    const uint32_t part = 0u;
    if constexpr (REGS == REGS_BINS) {
        regs = bin_regs_of<ALGO>(a, it.genome);                            // no table here: entries for bins_apply_kernel
        census = lds_regs;
    } else if constexpr (REGS == REGS_BYTES) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_regs != 0u) __builtin_trap();
        regs.p = p;
        census = lds_regs + a.nreg32;
    } else if constexpr (USE_LDS) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_regs != 0u) __builtin_trap();
        regs.base = lds_regs;
        census = lds_regs + a.nreg32;
    } else {
        regs.base = a.gregs + (uint64_t)(item - a.item_base) * a.nreg32;   // (tables of the items of this launch: lash_api.hip, global_run)           // zeroed by the host (hipMemsetAsync)
        census = lds_regs;
    }

    const uint32_t *__restrict__ w = a.words + gd.word_off;
    // A genome with a single record has no interior record boundary: its lanes read three always-zero words (one
    // L1-resident line) instead of streaming 1/8 B per base of break bitmap from HBM.  No branch, no second loop.
    const bool multi_rec = gd.format != 0u || gd.rec_end - gd.rec_begin > 1;
    // direct mode, all records of the genome equally long (a FASTQ read set; rec_uniform_kernel): record starts are the multiples
    // of RL — computed per lane and tile instead of read from a bitmap that then need not exist
    uint32_t RL = 0;
    if constexpr (DIRECT) {
        if (multi_rec && gd.format == 0u && a.nonuniform[it.genome] == 0u) RL = (uint32_t)(a.rec_off[gd.rec_begin + 1] - a.rec_off[gd.rec_begin]);
    }
    const bool use_bitmap = multi_rec && RL == 0u;
    const uint32_t *__restrict__ bk = use_bitmap ? (DIRECT ? a.brk_bytes : a.brk) + gd.brk_off : a.zero_words;
    const uint8_t *__restrict__ gseq = DIRECT ? a.seq + gd.byte_off : nullptr;
    uint32_t *const dirty = DIRECT ? a.dirty + it.genome : nullptr;
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
    const CodeTabs ctabs{a.lay.code_lo, a.lay.code_hi};
    constexpr bool K21 = KMODE == KM_GT16 && DIRECT && !DEFER && (REGS == REGS_LDS || REGS == REGS_BYTES);   // kernels with a k = 21 body of their own
    uint32_t my_kmers = 0;
    const uint32_t lane = threadIdx.x & 63u;
    SigQueue sigq;                                                          // DEFER: the lanes' stacks live in the wave's staging area
    sigq_init(sigq, (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.stage_off + (threadIdx.x >> 6) * a.stage_stride)), a.sigq_depth, lane);
    if constexpr (Regs::QUEUED)                                             // (the byte tables' stacks: same place, same layout)
        regs.queue_init((uint32_t)__builtin_amdgcn_readfirstlane((int)(a.stage_off + (threadIdx.x >> 6) * a.stage_stride)), a.sigq_depth, lane);
    uint32_t tiles_dense = 0;                                               // direct mode, per wave: how many of its tiles held deleted bytes
    bool judged = false;                                                    // ... and whether it has already voted to hand the genome over

    // One tile = blockDim.x * 4 words.  The next tile's words and break bits are loaded into registers before the
    // current tile is hashed (about 10k cycles of VALU work per tile cover the HBM latency).
    // direct mode keeps the raw bytes in registers until they are hashed: 4 x 16 own bytes + 1 (2) look-ahead chunks
    struct TileRegs { uint4 q; uint4 a1, a2, a3, la, lb; uint4 b; uint32_t c4, c5, dflag; };
    const uint32_t step = blockDim.x * SKETCH_WORDS_PER_THREAD;
    auto tile_active = [&](uint32_t tile) {
        const uint32_t w0 = tile + threadIdx.x * SKETCH_WORDS_PER_THREAD;
        return tile < it.word_end && w0 < it.word_end && w0 < nk_words;
    };
    // Loads are unconditional (inactive lanes and the prefetch past the last tile read a clamped, in-bounds
    // address): a predicated load sits in an exec-masked block and hipcc then waits for it at the block's end,
    // which would expose the HBM latency once per tile.
    const uint32_t w_last = it.word_end - SKETCH_WORDS_PER_THREAD;        // slices are >= 4 words, multiples of 4
    auto direct_fast = [&](uint32_t w0) { return w0 < fast_words; };              // all 96 bytes inside the genome
    auto tile_load = [&](uint32_t tile, TileRegs &t) {
        uint32_t w0 = tile + threadIdx.x * SKETCH_WORDS_PER_THREAD;
        w0 = w0 < w_last ? w0 : w_last;
        if constexpr (DIRECT) {
            const uint8_t *src = direct_fast(w0) ? gseq + (uint64_t)w0 * 16 : a.safe;   // any alignment: byte_off is arbitrary
            t.q = load16_any(src); t.a1 = load16_any(src + 16); t.a2 = load16_any(src + 32); t.a3 = load16_any(src + 48);
            t.la = load16_any(src + 64);
            if constexpr (KMODE == KM_GT16) t.lb = load16_any(src + 80);
            t.dflag = __hip_atomic_load(dirty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            t.q = *reinterpret_cast<const uint4 *>(w + w0);               // 64 bases, 16 B per lane, coalesced
            t.c4 = w[w0 + 4];                                             // look-ahead (same or next cache line)
            t.c5 = (KMODE == KM_GT16) ? w[w0 + 5] : 0u;
        }
        const uint32_t bi = use_bitmap ? w0 >> 1 : 0u;                    // (w0 * 16) / 32
        // three words are needed; FOUR are loaded: a dwordx3 lands in three consecutive registers that are not the ones the loop carries,
        // hipcc then copies two of them — and waits for every load of the prefetch just issued to do so, at the top of each tile
        // (every genome's bitmap, and the zero words, have the fourth word: lash_api.hip, bo += ... + 4)
        t.b = load16_any(reinterpret_cast<const uint8_t *>(bk + bi));
    };
    TileRegs nxt;
    tile_load(it.word_begin, nxt);
    // the table is cleared while the first tile's loads are in flight; a raw barrier, because __syncthreads() would also
    // drain vmcnt and with it those loads
    if constexpr (USE_LDS && REGS != REGS_BINS) {
        for (uint32_t i = threadIdx.x; i < a.nreg32; i += blockDim.x) lds_regs[i] = ALGO == 2 ? 0u : RANK_EMPTY;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef LASH_ABL_PROLOGUE_ONLY   // timing-only diagnostic builds (tools/build_variant_lib.sh): results are wrong by construction
    if (a.k != 99) return;
#endif
    for (uint32_t tile = it.word_begin; tile < it.word_end; tile += step) {
        const uint32_t w0 = tile + threadIdx.x * SKETCH_WORDS_PER_THREAD;
        const uint64_t pos0 = (uint64_t)w0 * 16;
        const bool active = tile_active(tile);
        const TileRegs cur = nxt;
        asm volatile("" ::"v"(cur.b.w));                                // (keeps the bitmap load four words wide: see tile_load)
        // The next tile's loads are issued AFTER this tile's bytes have been converted (and its dirt, if any, dealt with), right
        // before the hashing — not here.  Issued here, hipcc's wait for this tile's registers at the conversion is a vmcnt(0) that
        // also waits for the loads just issued (the counter is in order, and the loop's other paths make it give up counting): the
        // prefetch never overlapped anything, from round 1 on ("prefetch on/off is neutral").  Converted first, the only loads
        // outstanding at that wait are a whole tile of hashing old; and the raw bytes of two tiles are never live together.
        // A wave with no lane inside the slice has nothing to hash (a 10 kbp genome fills 2.5 of a workgroup's 8 waves).
        // Letting it run the masked body is worse than wasted issue slots: its lanes would all hash the same all-zero
        // words and hit ONE LDS address with 64-way serialized atomics (46 us per small genome instead of ~10).
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) { tile_load(tile + step, nxt); continue; }

        // inactive lanes of an active wave: distinct garbage words (their updates are masked to no-ops), for the same reason
        const uint32_t junk = (threadIdx.x + 1u) * 0x9E3779B1u;
        uint32_t c0 = junk, c1 = ~junk, c2 = junk, c3 = ~junk, c4 = junk, c5 = ~junk;
        uint64_t kv = 0;
        if constexpr (DIRECT) {
            // another workgroup (or an earlier tile) met a byte outside ACGT: this genome goes to the fallback path
            if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)cur.dflag) != 0, 0)) break;
            uint32_t bad = 0;
            if (active) {
                if (__builtin_expect(direct_fast(w0), 1)) {
                    c0 = ascii16_to_word(cur.q, bad, ctabs); c1 = ascii16_to_word(cur.a1, bad, ctabs); c2 = ascii16_to_word(cur.a2, bad, ctabs);
                    c3 = ascii16_to_word(cur.a3, bad, ctabs); c4 = ascii16_to_word(cur.la, bad, ctabs);
                    if constexpr (KMODE == KM_GT16) c5 = ascii16_to_word(cur.lb, bad, ctabs);
                } else {
                    const TailWords t = ascii96_tail(gseq, (uint64_t)w0 * 16, L, ctabs);
                    c0 = t.w[0]; c1 = t.w[1]; c2 = t.w[2]; c3 = t.w[3]; c4 = t.w[4];
                    if constexpr (KMODE == KM_GT16) c5 = t.w[5];
                    bad |= t.bad;
                }
                if (RL) kv = uniform_valid_mask((uint32_t)pos0, RL, (uint32_t)nk, k);
                else kv = kmer_valid_mask(cur.b.x, cur.b.y, cur.b.z, (uint32_t)pos0, (uint32_t)nk, k);
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad != 0u) != 0ull, 0)) {      // (unlikely: keeps its spills out of the clean path)
                // bytes outside the alphabet in this wave's tile.  Sparse dirt (an IUPAC code, an N in a read): the lanes that own
                // junction k-mers walk them, below.  Dense dirt — more than a quarter of the 4 KiB deleted, or a junction k-mer whose
                // bases lie beyond its lane's 96 bytes (the flank of a gap or of a soft-masked block): the wave compacts the tile in
                // LDS and hashes the survivors (dense_tile).
                const bool raw_ok = active && direct_fast(w0);                 // cur.q .. cur.a3 are this lane's own 64 bytes
                {
                    // the inside of a soft-masked block or of a gap: every active lane sees nothing but lower case / N -> nothing to do
                    const bool gone = raw_ok && (hopeless_bits(cur.q) & hopeless_bits(cur.a1) & hopeless_bits(cur.a2) & hopeless_bits(cur.a3) &
                                                 0x20202020u) == 0x20202020u;
                    const uint64_t act = __builtin_amdgcn_ballot_w64(active);
                    if (__builtin_amdgcn_ballot_w64(gone) == act) {
                        // (raw_ok lanes own exactly their 64 bytes: the genome's tail is at least 32 bytes away)
                        if ((threadIdx.x & 63) == 0 && part == 0u) atomicAdd(a.ndel + it.genome, 64u * (uint32_t)__builtin_popcountll(act));
                        tile_load(tile + step, nxt);
                        continue;
                    }
                }
                // A genome whose tiles need this often — walked or compacted: a soft-masked assembly, a read set with an N in every
                // hundredth read — is better off in stream_sketch_kernel, which filters as it reads (tools/dirty_rate.py).
                // The wave's own tiles are a sample of the genome: from one dirty tile in eight (and at least four) it hands the
                // genome over; the occasional gap or IUPAC code stays here and costs no second pass.  (Until round 3 a per-genome
                // counter in memory held a budget of walked tiles: one returning atomic per dirty tile on ONE address, which for a
                // single 3 Gbp read set with 20 000 Ns serialised in the L2 and cost more than the walks — 2.9 -> 4.4 ms.)
                tiles_dense = (uint32_t)__builtin_amdgcn_readfirstlane((int)tiles_dense) + 1u;
                if (tiles_dense >= 4u && tiles_dense * 8u > (tile - it.word_begin) / step + 1u && !judged) {
                    // this wave's verdict, once; the genome goes when enough of its waves agree (GenomeDesc::handover: one for a
                    // genome of a few items, 1 in 32 for a read set of thousands — there SOME wave always meets four dirty tiles early)
                    judged = true;
                    if ((threadIdx.x & 63) == 0 && atomicAdd(a.nslow + it.genome, 1u) + 1u >= gd.handover)
                        atomicOr(dirty, 1u);                                   // (seen by every wave of the genome at its next tile load)
                }
                // a quarter of the lanes met deleted bytes: dense without looking closer
                bool dense = __builtin_popcountll(__builtin_amdgcn_ballot_w64(bad != 0u)) >= 16;
                uint32_t nd = 0, wave_nd = 0;
                uint64_t junc = 0, jstarts = 0;
                if (!dense) {
                    bool far = false;
                    if (active && bad != 0u) {
                        const InvMask im = lane_inv_mask(gseq, pos0, L);
                        const uint64_t W = window_or(im.lo, im.hi, k);            // windows that hold a deleted byte
                        const uint64_t in_gen = L - pos0 >= 64 ? ~0ull : ((1ull << (L - pos0)) - 1ull);
                        junc = ~im.lo & W & in_gen;                                // surviving first base, broken window
                        jstarts = junc ? (~im.lo & in_gen & ~((1ull << __builtin_ctzll(junc)) - 1ull)) : 0ull;
                        kv &= ~W;
                        // deleted bytes this lane accounts for: its own 64, and the genome's last <= k-1 bytes when no lane starts there
                        const uint64_t own = pos0 + 64 >= nk ? L - pos0 : 64;
                        nd = (uint32_t)__builtin_popcountll(im.lo & in_gen) +
                             (own > 64 ? (uint32_t)__builtin_popcount(im.hi & ((own >= 96 ? 0u : (1u << (own - 64))) - 1u)) : 0u);
                        if (junc) {
                            // the last junction start needs k-1 survivors after it; are they among the lane's 96 bytes?
                            const uint32_t bl = 63u - (uint32_t)__builtin_clzll(junc);
                            const uint64_t in_gen_hi = L - pos0 >= 96 ? 0xFFFFFFFFull : (L - pos0 > 64 ? ((1ull << (L - pos0 - 64)) - 1ull) : 0ull);
                            const uint64_t after = bl == 63u ? 0ull : (~im.lo & in_gen) >> (bl + 1u);
                            far = (uint32_t)__builtin_popcountll(after) + (uint32_t)__builtin_popcountll(~(uint64_t)im.hi & in_gen_hi) < (uint32_t)k - 1u &&
                                  pos0 + 96 < L;                               // (the genome's end cuts the walk short anyway)
                        }
                    }
                    wave_nd = wave_sum(nd);
                    dense = wave_nd > 1024u || __builtin_amdgcn_ballot_w64(far) != 0ull;
                }
                if (dense) {
                    const uint64_t P0 = 16ull * (tile + (threadIdx.x & ~63u) * SKETCH_WORDS_PER_THREAD);
                    uint64_t E = P0 + 4096 < 16ull * it.word_end ? P0 + 4096 : 16ull * it.word_end;
                    E = E < L ? E : L;
                    const uint32_t stage_b = a.stage_off + (threadIdx.x >> 6) * a.stage_stride;
                    if constexpr (DEFER) sigq_drain<true>(regs, kp.bitflip, p, sigq);           // (the staging area is about to be used)
                    if constexpr (Regs::QUEUED) regs.template drain<true>();
                    my_kmers += wave_sum(dense_tile<ALGO, KMODE, XLOW, SlowRegs>(regs, kp, gseq, L, P0, E, use_bitmap ? bk : nullptr, RL, k, cmask, ctabs,
                                                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)stage_b), dirty,
                                                                            part == 0u ? a.ndel + it.genome : nullptr, raw_ok, cur.q, cur.a1, cur.a2, cur.a3));
                    tile_load(tile + step, nxt);
                    continue;
                }
                if (wave_nd && part == 0u && (threadIdx.x & 63) == 0) atomicAdd(a.ndel + it.genome, wave_nd);   // (one per wave; the passes of a partitioned table see the same bytes)
                uint32_t walked = 0;
                if (junc)                                                      // (here, not after the hashing: nothing of it stays live)
                    walked = junction_walk<ALGO, XLOW, SlowRegs>(regs, gseq, L, pos0, junc, jstarts, use_bitmap ? bk : nullptr, RL, k, kp.bitflip, p,
                                                             cmask, ctabs, dirty);
                my_kmers += wave_sum(walked);
            }
        } else if (active) {
            c0 = cur.q.x; c1 = cur.q.y; c2 = cur.q.z; c3 = cur.q.w; c4 = cur.c4; c5 = cur.c5;
            kv = kmer_valid_mask(cur.b.x, cur.b.y, cur.b.z, (uint32_t)pos0, (uint32_t)nk, k);
        }
        tile_load(tile + step, nxt);                                    // (see the top of the loop)
        // wave-uniform: every lane of this wave has 64 real k-mers -> no per-k-mer masking at all
        const bool all_valid = __builtin_amdgcn_ballot_w64(kv != ~0ull) == 0ull;
        // the k-mer census is kept per WAVE in a scalar register (a vector register less across the hashing: the k > 16 kernels sit
        // at their 128-register cap and spilled this one on every tile): 4 096 for a full tile, a wave sum otherwise
        if (all_valid) my_kmers += 4096u;
        else my_kmers += wave_sum((uint32_t)__builtin_popcountll(kv));

        uint32_t r0 = rcword(c0, cmask), r1 = rcword(c1, cmask), r2 = (KMODE == KM_GT16) ? rcword(c2, cmask) : 0u;
        if constexpr (REGS == REGS_BINS) regs.mode = 1u;                   // the word loop pushes without branches (BinRegs::mode)
#pragma unroll LASH_WORD_UNROLL
        for (int wi = 0; wi < SKETCH_WORDS_PER_THREAD; ++wi) {
            uint32_t z;
            if constexpr (DEFER) {
                z = 0xFFFFFFFFu;                                               // (nothing to re-run: the full update does that itself)
                if (all_valid) process_word_defer<KMODE, false>(regs, kp, c0, c1, c2, r0, r1, r2, 0u, sigq);
                else {
                    uint32_t kvw = (uint32_t)kv;
                    asm volatile("" : "+v"(kvw));
                    process_word_defer<KMODE, true>(regs, kp, c0, c1, c2, r0, r1, r2, kvw, sigq);
                }
            } else if (K21 && k == 21) {
                // k = 21 (BASELINE configs[2]; the fields of the 64-bit window at compile-time places: canon_gt16<21>)
                if (all_valid) z = process_word<ALGO, KMODE, XLOW, false, true, Regs, K21 ? 21 : 0>(regs, kp, c0, c1, c2, r0, r1, r2, 0u);
                else {
                    uint32_t kvw = (uint32_t)kv;
                    asm volatile("" : "+v"(kvw));
                    z = process_word<ALGO, KMODE, XLOW, true, true, Regs, K21 ? 21 : 0>(regs, kp, c0, c1, c2, r0, r1, r2, kvw);
                }
            } else if (all_valid) {
                z = process_word<ALGO, KMODE, XLOW, false, true>(regs, kp, c0, c1, c2, r0, r1, r2, 0u);
            } else {
                // the masks are taken from an opaque copy so that hipcc cannot hoist the 16 v_bfe_i32 above the
                // branch, where the (usual) all-valid path would pay for them too
                uint32_t kvw = (uint32_t)kv;
                asm volatile("" : "+v"(kvw));
                z = process_word<ALGO, KMODE, XLOW, true, true>(regs, kp, c0, c1, c2, r0, r1, r2, kvw);
            }
            // FAST forms return a word whose smallness flags "rank not decided by the bits looked at": exact re-run
            // (HMH/x-high looks at 18 bits -> 2^-18 per k-mer; the others at 32 bits -> 2^-32)
            constexpr uint32_t Z_REDO = z_redo<ALGO, Regs>();
            if (z <= Z_REDO) {
                uint32_t kvw = (uint32_t)kv;
                asm volatile("" : "+v"(kvw));
                (void)process_word<ALGO, KMODE, XLOW, true, false>(regs, kp, c0, c1, c2, r0, r1, r2, kvw);
            }
            if constexpr (REGS == REGS_BINS) {
                if (regs.overflowed()) {                                       // a staging row ran full: the word again, straight to the fallback table
                    regs.mode = 2u;
                    uint32_t kvw = (uint32_t)kv;
                    asm volatile("" : "+v"(kvw));
                    (void)process_word<ALGO, KMODE, XLOW, true, false>(regs, kp, c0, c1, c2, r0, r1, r2, kvw);
                    regs.mode = 1u;
                }
                if ((((uint32_t)wi + 1u) & (a.bin_flush_words - 1u)) == 0u) regs.flush(lane);   // the staged entries leave the wave's rows (every word, or every 2nd / 4th)
            }
            // rotate the window by one word
            c0 = c1; c1 = c2; c2 = c3; c3 = c4; c4 = c5; c5 = 0;
            r0 = r1;
            if constexpr (KMODE == KM_GT16) { r1 = r2; r2 = rcword(c2, cmask); } else { r1 = rcword(c1, cmask); }
            kv >>= 16;
        }
        if constexpr (REGS == REGS_BINS) regs.mode = 0u;
    }

    if constexpr (DEFER) sigq_drain<true>(regs, kp.bitflip, p, sigq);
    if constexpr (Regs::QUEUED) regs.template drain<true>();
#ifdef LASH_ABL_NO_FLUSH
    if (a.k != 99) return;
#endif
    finish_item<ALGO, REGS, Regs>(a, it, regs, census, part, my_kmers, p, item);
#ifdef LASH_ITEM_TRACE_BUILD
    if (a.item_trace && threadIdx.x == 0) a.item_trace[4ull * item + 1] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------------------
// stream_sketch_kernel — the genomes the direct pass gave up (dense_tile's fallback ratio, a gap beyond DENSE_SCAN_MAX), still
// from their ASCII bytes and in ONE pass: no 2-bit round trip through HBM, no second kernel.
//
// Same work items as the direct pass.  Each wave takes a CONTIGUOUS part of the item's bytes and walks it 2 KiB at a time:
// classify (32 bytes per lane), DPP prefix sum, survivors appended to a wave-private ring in LDS (the staging area of
// dense_tile: 260 words of 2-bit codes + 130 words of record-start bits) — and whenever the ring holds 2 048 bases plus the k-1
// that follow them, a batch of 2 048 k-mer starts is hashed: two packed words per lane, every lane busy, the clean path's
// process_word.  Because a wave reads its bytes in order there is nothing to look ahead for or to round up: the k-mers of a
// wave are those whose first base is a surviving byte of its part, so after its last byte it keeps reading until k-1 more
// bases have survived (or the record / genome ends), then hashes what is left under a mask.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t RING_W = 260, RING_BW = 130;          // code words / record-start words (4 160 bases); both fit DENSE_STAGE_WORDS
constexpr uint32_t STREAM_CHUNK = 2048, STREAM_BATCH = 2048;
static_assert(RING_W <= DENSE_STAGE_CODE_WORDS && RING_BW <= DENSE_STAGE_BRK_WORDS, "the ring lives in dense_tile's staging area");

// DEFER (round 5): HyperMinHash genomes of long work items — a soft-masked assembly is exactly that — run the
// filter of process_word_defer here too: the rank half of the hash for every k-mer, the signature half only for the few whose rank
// can still win their bucket (threshold words, per-lane stacks: LdsThrRegs, SigQueue).  The stacks sit BEHIND the waves' rings, and
// the launch is 1 024 threads (one workgroup per CU, sixteen waves: the waves of this kernel never wait for each other between the
// first and the last barrier, so one large workgroup loses nothing to two small ones and has room for both areas).
template <int ALGO, int KMODE, bool XLOW, int REGS, bool DEFER = false>
__global__ void __launch_bounds__(1024) stream_sketch_kernel(SketchArgs a)
{
    static_assert(!DEFER || (ALGO == 0 && REGS == REGS_LDS), "deferred signatures: HyperMinHash, one LDS table");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_regs[];
    const uint32_t slot = xcd_skewed(blockIdx.x, gridDim.x);
    const uint32_t item = a.item_order ? a.item_order[slot] : slot + a.item_base;
    const WorkItem it = a.items[item];
    if (a.dirty[it.genome] == 0u) return;                                  // only the genomes the direct pass gave up
    const GenomeDesc gd = a.genomes[it.genome];
    const uint64_t L = gd.byte_len;
    const int k = a.k, p = a.p;
    const uint64_t nkg = L >= (uint64_t)k ? L - (uint64_t)k + 1 : 0;
    if ((uint64_t)it.word_begin * 16 >= nkg) {                            // (as in sketch_kernel: the item holds no k-mer start)
        if ((it.slice & ITEM_SOLE) && !a.accumulate) {
            uint8_t *img = a.images + (uint64_t)it.genome * a.image_bytes;
            for (uint64_t i = threadIdx.x; i < a.image_bytes; i += blockDim.x) img[i] = 0;
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint64_t n_regs = ALGO == 0 ? HMH_M : (1ull << p);
                write_header(img, a.lay.hdr_tpl, a.alpha_bits, n_regs, n_regs, (double)n_regs, ALGO == 0 ? HMH_P : p);
            }
        }
        return;
    }
    constexpr bool USE_LDS = REGS != REGS_GLOBAL;
    using Regs = typename std::conditional<DEFER, typename std::conditional<XLOW, LdsThrRegsX, LdsThrRegs>::type,
                                           typename std::conditional<REGS == REGS_LDS, LdsRegs,
                                           typename std::conditional<REGS == REGS_GLOBAL, GlobalRegs,
                                           typename std::conditional<REGS == REGS_BINS, BinRegs, LdsByteRegs>::type>::type>::type>::type;
    Regs regs;
    uint32_t *census;
    const uint32_t part = 0u;
    if constexpr (REGS == REGS_BINS) {
        regs = bin_regs_of<ALGO>(a, it.genome);
        census = lds_regs;
    } else if constexpr (REGS == REGS_BYTES) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_regs != 0u) __builtin_trap();
        regs.p = p;
        census = lds_regs + a.nreg32;
        for (uint32_t i = threadIdx.x; i < a.nreg32; i += blockDim.x) lds_regs[i] = ALGO == 2 ? 0u : RANK_EMPTY;
    } else if constexpr (USE_LDS) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_regs != 0u) __builtin_trap();
        regs.base = lds_regs;
        census = lds_regs + a.nreg32;
        for (uint32_t i = threadIdx.x; i < a.nreg32; i += blockDim.x) lds_regs[i] = ALGO == 2 ? 0u : RANK_EMPTY;
    } else {
        regs.base = a.gregs + (uint64_t)(item - a.item_base) * a.nreg32;   // (tables of the items of this launch: lash_api.hip, global_run)
        census = lds_regs;
    }
    const bool multi_rec = gd.rec_end - gd.rec_begin > 1;
    uint32_t RL = 0;
    if (multi_rec && a.nonuniform[it.genome] == 0u) RL = (uint32_t)(a.rec_off[gd.rec_begin + 1] - a.rec_off[gd.rec_begin]);
    const uint32_t *__restrict__ bk = (multi_rec && RL == 0u) ? a.brk_bytes + gd.brk_off : nullptr;
    const bool breaks = bk != nullptr || RL != 0u;
    const uint8_t *__restrict__ gseq = a.seq + gd.byte_off;
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

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t stage_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.stage_off + wave * (DENSE_STAGE_WORDS * 4u)));
    const uint32_t brk_b = stage_b + 4u * DENSE_STAGE_CODE_WORDS;
    lds_u32 *const ring = (lds_u32 *)(uintptr_t)stage_b;
    lds_u32 *const bring = (lds_u32 *)(uintptr_t)brk_b;
    SigQueue sigq;                                                          // DEFER: the lanes' stacks, behind every wave's ring
    sigq_init(sigq, (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.stage_off + n_waves * (DENSE_STAGE_WORDS * 4u) + wave * (64u * a.sigq_depth * 4u))),
              a.sigq_depth, lane);
    for (uint32_t i = lane; i < DENSE_STAGE_WORDS; i += 64u) ring[i] = 0;
    __syncthreads();                                                       // the register table and every ring are clear

    // this wave's part of the item: [ws, we) in genome bytes, 2 KiB-aligned relative to the item's first byte
    const uint64_t B0 = 16ull * it.word_begin, B1 = 16ull * it.word_end < L ? 16ull * it.word_end : L;
    const uint64_t span = (((B1 - B0 + n_waves - 1) / n_waves) + STREAM_CHUNK - 1) & ~(uint64_t)(STREAM_CHUNK - 1);
    const uint64_t ws = B0 + (uint64_t)wave * span, we = ws + span < B1 ? ws + span : B1;
    uint32_t my_kmers = 0;
    if (ws < B1) {
        const uint32_t want = (uint32_t)k - 1u;                            // bases a k-mer needs after its first
        uint32_t head_w = 0, have = 0, own_left = 0, la_have = 0, own_seen = 0;   // wave-uniform ring state (bases after head; ...)
        bool pend = false, stop = false;
        // one batch of `nstarts` k-mer starts (<= 2 048) from the ring's head; the ring must hold nstarts + k - 1 bases (or end there)
        auto hash_batch = [&](uint32_t nstarts) {
            const uint32_t pos0 = 32u * lane;
            const bool active = pos0 < nstarts;
            const uint32_t junk = (threadIdx.x + 1u) * 0x9E3779B1u;
            uint32_t c0 = junk, c1 = ~junk, c2 = junk, c3 = ~junk;
            uint32_t kvw = 0;
            auto rw = [&](uint32_t j) { uint32_t w = head_w + 2u * lane + j; w = w >= RING_W ? w - RING_W : w; return w; };
            const uint32_t w0 = rw(0), w1 = rw(1);
            if (active) {
                c0 = ring[w0]; c1 = ring[w1]; c2 = ring[rw(2)]; c3 = ring[rw(3)];
                uint32_t b0 = 0, b1 = 0;
                if (breaks) {
                    uint32_t bw = (head_w >> 1) + lane; bw = bw >= RING_BW ? bw - RING_BW : bw;
                    const uint32_t bw1 = bw + 1u == RING_BW ? 0u : bw + 1u;
                    b0 = bring[bw]; b1 = bring[bw1];
                }
                kvw = (uint32_t)kmer_valid_mask(b0, b1, 0u, pos0, nstarts, k);
            }
            const bool all_valid = __builtin_amdgcn_ballot_w64(kvw != 0xFFFFFFFFu) == 0ull;
            my_kmers += all_valid ? 2048u : wave_sum((uint32_t)__builtin_popcount(kvw));
            // the batch's words are free again (appends OR into zeroed words); the two after it belong to the next batch
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            ring[w0] = 0; ring[w1] = 0;
            if (breaks) { uint32_t bw = (head_w >> 1) + lane; bw = bw >= RING_BW ? bw - RING_BW : bw; bring[bw] = 0; }
            uint32_t r0 = rcword(c0, cmask), r1 = rcword(c1, cmask), r2 = (KMODE == KM_GT16) ? rcword(c2, cmask) : 0u;
#pragma unroll 1
            for (int wi = 0; wi < 2; ++wi) {
                uint32_t z;
                if constexpr (DEFER) {
                    z = 0xFFFFFFFFu;                                           // (nothing to re-run: the full update does that itself)
                    if (all_valid) process_word_defer<KMODE, false>(regs, kp, c0, c1, c2, r0, r1, r2, 0u, sigq);
                    else {
                        uint32_t m = kvw;
                        asm volatile("" : "+v"(m));
                        process_word_defer<KMODE, true>(regs, kp, c0, c1, c2, r0, r1, r2, m, sigq);
                    }
                } else if (all_valid) z = process_word<ALGO, KMODE, XLOW, false, true>(regs, kp, c0, c1, c2, r0, r1, r2, 0u);
                else {
                    uint32_t m = kvw;
                    asm volatile("" : "+v"(m));
                    z = process_word<ALGO, KMODE, XLOW, true, true>(regs, kp, c0, c1, c2, r0, r1, r2, m);
                }
                constexpr uint32_t Z_REDO = z_redo<ALGO, Regs>();
                if (z <= Z_REDO) {
                    uint32_t m = kvw;
                    asm volatile("" : "+v"(m));
                    (void)process_word<ALGO, KMODE, XLOW, true, false>(regs, kp, c0, c1, c2, r0, r1, r2, m);
                }
                if constexpr (REGS == REGS_BINS) regs.flush(lane);
                c0 = c1; c1 = c2; c2 = c3; c3 = 0;
                r0 = r1;
                if constexpr (KMODE == KM_GT16) { r1 = r2; r2 = rcword(c2, cmask); } else { r1 = rcword(c1, cmask); }
                kvw >>= 16;
            }
            head_w += STREAM_BATCH / 16u;
            head_w = head_w >= RING_W ? head_w - RING_W : head_w;
        };
        // the chunk's bytes, 32 per lane, asked for one chunk ahead (lanes whose 32 bytes are not all inside the genome read a
        // dummy line and fetch their bytes one by one when their turn comes)
        auto chunk_load = [&](uint64_t cp, uint4 &x0, uint4 &x1) {
            const uint64_t ca = cp + 32ull * lane;
            const uint8_t *src = ca + 32 <= L ? gseq + ca : a.safe;
            x0 = load16_any(src); x1 = load16_any(src + 16);
        };
        uint4 n0, n1;
        // past the wave's part, inside a long run of deleted bytes (a gap, a masked block): 8 KiB per round trip to the first chunk
        // that may hold a survivor or a record start — the workgroup's slot waits for this wave.  (pos: the chunk that held nothing)
        auto gap_skip = [&](uint64_t &pos) {
            if (pos >= we && !stop) {
                uint64_t np = pos + STREAM_CHUNK;
                for (;;) {
                    if (np + 4 * STREAM_CHUNK + 32 > L) break;              // near the genome's end: chunk by chunk
                    uint4 y[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        y[2 * j] = load16_any(gseq + np + (uint64_t)j * STREAM_CHUNK + 32ull * lane);
                        y[2 * j + 1] = load16_any(gseq + np + (uint64_t)j * STREAM_CHUNK + 32ull * lane + 16);
                    }
                    uint32_t first = 4;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        bool hit = (hopeless_bits(y[2 * j]) & hopeless_bits(y[2 * j + 1]) & 0x20202020u) != 0x20202020u;
                        if (breaks) {
                            const uint64_t aj = np + (uint64_t)j * STREAM_CHUNK + 32ull * lane;
                            hit = hit || (RL ? uniform_breaks((uint32_t)aj, RL, 32u).b0 != 0u : bk[aj >> 5] != 0u);
                        }
                        if (__builtin_amdgcn_ballot_w64(hit) != 0ull) first = (uint32_t)j;
                    }
                    np += (uint64_t)first * STREAM_CHUNK;
                    if (first < 4u) break;
                }
                if (np != pos + STREAM_CHUNK) { pos = np - STREAM_CHUNK; chunk_load(np, n0, n1); }
            }
        };
        bool was_gap = false;                                              // the last chunk held no survivor
        chunk_load(ws, n0, n1);
        for (uint64_t pos = ws; pos < L && !stop; pos += STREAM_CHUNK) {
            if (pos >= we && (own_left == 0u || la_have >= want)) break;    // nothing owned waits for more bases
            const uint64_t at = pos + 32ull * lane;
            // record starts among the lane's 32 positions (multi-record genomes: contigs, reads).  The bitmap word is asked for BEFORE the
            // next chunk's bytes: outstanding loads retire in order, so whoever waits for a load waits for every load issued before it —
            // with this one behind the prefetch, hipcc's wait for its register sat at the top of the hash loop as vmcnt(0) (also for
            // genomes without record starts: the register is overwritten there) and the prefetch never overlapped the hashing
            uint32_t rb = 0;
            if (breaks && at < L) rb = RL ? uniform_breaks((uint32_t)at, RL, 32u).b0 : bk[at >> 5];
            uint4 q0 = n0, q1 = n1;
            chunk_load(pos + STREAM_CHUNK, n0, n1);
            if (at + 32 > L) {
                uint32_t d[8];
                for (int i = 0; i < 8; ++i) d[i] = 0x4E4E4E4Eu;                  // 'N': beyond the genome nothing survives
                for (uint32_t i = 0; i < 32 && at + i < L; ++i) d[i >> 2] = (d[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((uint32_t)gseq[at + i] << (8 * (i & 3)));
                q0 = make_uint4(d[0], d[1], d[2], d[3]); q1 = make_uint4(d[4], d[5], d[6], d[7]);
            }
            // ---- classification in three steps: is anything deleted at all? (most chunks of most genomes: no) — is anything left?
            //      (the inside of a masked block: no) — else which bytes ----
            // Inside a masked block the next chunk is most likely masked too: ask THAT first and save the conversion (54 instructions
            // against 26; genomes with record starts keep the order: a start inside the run has to be carried, below)
            bool hop_known = false, gap = false;
            if (was_gap && !breaks) {
                hop_known = true;
                gap = __builtin_amdgcn_ballot_w64((hopeless_bits(q0) & hopeless_bits(q1) & 0x20202020u) != 0x20202020u) == 0ull;
            }
            uint32_t T = 0, own_t = 0;
            // the chunk's survivors into the ring; true: there are none
            auto chunk_body = [&]() -> bool {
                uint32_t bad = 0;
                const uint32_t cw0 = ascii16_to_word(q0, bad, ct), cw1 = ascii16_to_word(q1, bad, ct);
                // nothing deleted in an owned chunk (and no record start waiting for a survivor): every base goes to the ring as it is,
                // and a record start stays on its own base
                const uint64_t badm = __builtin_amdgcn_ballot_w64(bad != 0u);      // lanes that hold a deleted byte
                const bool chunk_clean = badm == 0ull && pos + STREAM_CHUNK <= we && !pend;
                if (chunk_clean) {
                    // 2 048 survivors, all owned: lane i's 32 bases go to ring position have + 32 i, one shift for all
                    const uint32_t rel = have + 32u * lane, sh = 2u * (rel & 15u);
                    uint32_t w = head_w + (rel >> 4); w = w >= RING_W ? w - RING_W : w;
                    const uint32_t w1 = w + 1u == RING_W ? 0u : w + 1u, w2 = w1 + 1u == RING_W ? 0u : w1 + 1u;
                    if (sh) {
                        lds_or(stage_b + 4u * w, cw0 >> sh);
                        lds_or(stage_b + 4u * w1, (cw0 << (32u - sh)) | (cw1 >> sh));
                        lds_or(stage_b + 4u * w2, cw1 << (32u - sh));
                    } else {
                        lds_or(stage_b + 4u * w, cw0);
                        lds_or(stage_b + 4u * w1, cw1);
                    }
                    if (breaks && __builtin_amdgcn_ballot_w64(rb != 0u) != 0ull && rb) {   // (reads: a dozen record starts per chunk)
                        uint32_t bw = (head_w >> 1) + (rel >> 5); bw = bw >= RING_BW ? bw - RING_BW : bw;
                        const uint32_t bwn = bw + 1u == RING_BW ? 0u : bw + 1u, bs = rel & 31u;
                        lds_or(brk_b + 4u * bw, rb << bs);
                        if (bs) lds_or(brk_b + 4u * bwn, rb >> (32u - bs));
                    }
                    T = STREAM_CHUNK; own_t = STREAM_CHUNK;
                } else {
                uint32_t v;
                // (a lane without a deleted byte answers the question for the chunk: only when every lane has one is it worth asking)
                if (!hop_known && badm == ~0ull && __builtin_amdgcn_ballot_w64((hopeless_bits(q0) & hopeless_bits(q1) & 0x20202020u) != 0x20202020u) == 0ull) v = 0u;
                else v = ~(inv16s(q0) | (inv16s(q1) << 16));                       // bit j: byte j survives (bytes past L are 'N')
                const uint32_t own_n = at >= we ? 0u : (we - at >= 32 ? 32u : (uint32_t)(we - at));
                const uint32_t ownmask = own_n >= 32u ? 0xFFFFFFFFu : ((1u << own_n) - 1u);
                // ---- record starts land on the first survivor at or after them (cf. dense_tile); one that lands BEYOND the part ends it ----
                uint32_t recv = 0;
                if (breaks) {
                    const uint32_t fill = ~v, rbd = rb & fill;
                    const bool gen = fill + rbd < rbd;
                    const uint64_t G = __builtin_amdgcn_ballot_w64(gen), Z = __builtin_amdgcn_ballot_w64(v == 0u);
                    const uint64_t below = (1ull << lane) - 1ull, nz = ~Z & below;
                    const uint64_t from = nz ? ~((1ull << (63 - __builtin_clzll(nz))) - 1ull) : ~0ull;
                    const bool pend_in = (G & below & from) != 0ull || (nz == 0ull && pend);
                    recv = ((fill + rbd + (pend_in ? 1u : 0u)) | rb) & v;
                    const uint64_t nzall = ~Z;
                    pend = nzall ? (G >> (63 - __builtin_clzll(nzall))) != 0ull : (pend || G != 0ull);
                    const uint32_t cut = recv & ~ownmask;                             // the next record opens here, past this wave's part
                    const uint64_t Cm = __builtin_amdgcn_ballot_w64(cut != 0u);
                    if (Cm) {
                        const uint32_t fl = (uint32_t)__builtin_ctzll(Cm);
                        if (lane > fl) v &= ownmask;
                        else if (lane == fl) v &= ownmask | ((1u << __builtin_ctz(cut)) - 1u);
                        stop = true;
                    }
                }
                const uint32_t n_all = (uint32_t)__builtin_popcount(v);
                const uint32_t off = wave_excl_scan(n_all, T);
                if (T == 0u) return true;                                          // nothing survives in this chunk
                own_t = pos + STREAM_CHUNK <= we ? T : (pos >= we ? 0u : wave_sum((uint32_t)__builtin_popcount(v & ownmask)));
                // ---- append: two 16-byte groups per lane at ring position have + off ----
                {
                    uint32_t rel = have + off;
                    // every lane's survivors one run of neighbours per group (soft-masked blocks, gap edges)?  Then compaction is a shift
                    const bool runs = __builtin_amdgcn_ballot_w64(!(is_run16(v & 0xFFFFu) && is_run16(v >> 16))) == 0ull;
    #pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const uint32_t m = (v >> (16 * c)) & 0xFFFFu;
                        uint32_t cb;
                        const uint32_t bits = runs ? compact16_run(c ? cw1 : cw0, m, (recv >> (16 * c)) & 0xFFFFu, cb)
                                                   : compact16(c ? cw1 : cw0, m, (recv >> (16 * c)) & 0xFFFFu, cb);
                        const uint32_t n = (uint32_t)__builtin_popcount(m);
                        if (n) {
                            uint32_t w = head_w + (rel >> 4); w = w >= RING_W ? w - RING_W : w;
                            const uint32_t wn = w + 1u == RING_W ? 0u : w + 1u, sh = 2u * (rel & 15u);
                            lds_or(stage_b + 4u * w, bits >> sh);
                            if (sh && (rel & 15u) + n > 16u) lds_or(stage_b + 4u * wn, bits << (32u - sh));
                            if (breaks && cb) {
                                uint32_t bw = (head_w >> 1) + (rel >> 5); bw = bw >= RING_BW ? bw - RING_BW : bw;
                                const uint32_t bwn = bw + 1u == RING_BW ? 0u : bw + 1u, bs = rel & 31u;
                                lds_or(brk_b + 4u * bw, cb << bs);
                                if (bs && bs + n > 32u) lds_or(brk_b + 4u * bwn, cb >> (32u - bs));
                            }
                        }
                        rel += n;
                    }
                }
                }
                return false;
            };
            if (!gap) gap = chunk_body();
            if (gap) {
                was_gap = true;
                gap_skip(pos);
                continue;
            }
            was_gap = false;
            have += T; own_left += own_t; own_seen += own_t; la_have += T - own_t;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            while (have >= STREAM_BATCH + want && own_left > 0u) {
                const uint32_t ns = own_left < STREAM_BATCH ? own_left : STREAM_BATCH;
                hash_batch(ns);
                have -= STREAM_BATCH; own_left -= ns;
                if (ns < STREAM_BATCH) break;                                  // (the owned starts are done; what is left in the ring is look-ahead)
            }
        }
        // ---- what is left: starts that have their k-1 followers, in batches, the rest of the ring behind them ----
        while (own_left > 0u) {
            const uint32_t room = have >= (uint32_t)k ? have - want : 0u;      // starts whose whole window is in the ring
            uint32_t ns = own_left < STREAM_BATCH ? own_left : STREAM_BATCH;
            ns = ns < room ? ns : room;
            if (ns == 0u) break;
            hash_batch(ns);
            have = have > STREAM_BATCH ? have - STREAM_BATCH : 0u;
            own_left -= ns;
            if (ns < STREAM_BATCH) break;
        }
        // surviving bases of this wave's part (lash_timing::bases_last)
        if (lane == 0 && part == 0u && a.ndel2) atomicAdd(a.ndel2 + it.genome, (uint32_t)(we - ws) - own_seen);
    }
    if constexpr (DEFER) sigq_drain<true>(regs, kp.bitflip, p, sigq);
    finish_item<ALGO, REGS, Regs>(a, it, regs, census, part, my_kmers, p, item);
}

template <int ALGO, int KMODE, bool XLOW, int REGS>
static hipError_t launch_stream_one(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream)
{
    auto kern = stream_sketch_kernel<ALGO, KMODE, XLOW, REGS>;
    uint32_t threads = plan.threads, stacks = 0;
    SketchArgs a = args;
    if constexpr (ALGO == 0 && REGS == REGS_LDS) {
        if (plan.defer) {                                                  // (see the kernel: 1 024 threads, the lanes' stacks behind the rings)
            kern = stream_sketch_kernel<ALGO, KMODE, XLOW, REGS, true>;
            threads = 1024u;
            stacks = (threads / 64u) * 64u * plan.sigq_depth * 4u;
        }
    }
    a.stage_off = plan.lds_bytes;
    a.sigq_depth = plan.sigq_depth;
    a.bin_lds_off = plan.lds_bytes + (threads / 64u) * (DENSE_STAGE_WORDS * 4u) + stacks;
    a.bin_wave_bytes = sketch_bin_wave_bytes(plan);
    const uint32_t lds = a.bin_lds_off + (threads / 64u) * a.bin_wave_bytes;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_items), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

template <int ALGO, bool XLOW>
static hipError_t launch_stream_kmode(const SketchPlan &plan, const SketchArgs &args, uint32_t n, hipStream_t s)
{
    const int km = plan.k == 16 ? KM_16 : plan.k < 16 ? KM_LT16 : KM_GT16;
    if constexpr (ALGO != 0) {
        if (plan.bytes) {
            if (km == KM_16) return launch_stream_one<ALGO, KM_16, XLOW, REGS_BYTES>(plan, args, n, s);
            if (km == KM_LT16) return launch_stream_one<ALGO, KM_LT16, XLOW, REGS_BYTES>(plan, args, n, s);
            return launch_stream_one<ALGO, KM_GT16, XLOW, REGS_BYTES>(plan, args, n, s);
        }
        if (plan.bins) {
            if (km == KM_16) return launch_stream_one<ALGO, KM_16, XLOW, REGS_BINS>(plan, args, n, s);
            if (km == KM_LT16) return launch_stream_one<ALGO, KM_LT16, XLOW, REGS_BINS>(plan, args, n, s);
            return launch_stream_one<ALGO, KM_GT16, XLOW, REGS_BINS>(plan, args, n, s);
        }
    }
    if (plan.use_lds) {
        if (km == KM_16) return launch_stream_one<ALGO, KM_16, XLOW, REGS_LDS>(plan, args, n, s);
        if (km == KM_LT16) return launch_stream_one<ALGO, KM_LT16, XLOW, REGS_LDS>(plan, args, n, s);
        return launch_stream_one<ALGO, KM_GT16, XLOW, REGS_LDS>(plan, args, n, s);
    }
    if constexpr (ALGO == 2) {
        if (km == KM_16) return launch_stream_one<ALGO, KM_16, XLOW, REGS_GLOBAL>(plan, args, n, s);
        if (km == KM_LT16) return launch_stream_one<ALGO, KM_LT16, XLOW, REGS_GLOBAL>(plan, args, n, s);
        return launch_stream_one<ALGO, KM_GT16, XLOW, REGS_GLOBAL>(plan, args, n, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_sketch_stream(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream)
{
    if (n_items == 0) return hipSuccess;
    switch (plan.algo) {
    case 0: return plan.variant ? launch_stream_kmode<0, true>(plan, args, n_items, stream) : launch_stream_kmode<0, false>(plan, args, n_items, stream);
    case 1: return plan.variant ? launch_stream_kmode<1, true>(plan, args, n_items, stream) : launch_stream_kmode<1, false>(plan, args, n_items, stream);
    case 2: return launch_stream_kmode<2, false>(plan, args, n_items, stream);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------------------
// amino-acid sketches: the `aa` branch of sketch_files (/root/reference/src/utils.rs:511-563), unreachable in the reference
// (main.rs:198 hard-wires aa = false).  Per record: upper-case (:521), skip when the RAW length is below k (:523-525), delete every
// byte outside the 20 residue letters (filter_out_a, utils.rs:43-55), 5 bits per residue in the order "ACDEFGHIKLMNPQRSTVWY"
// (kmerutils aautils [UNPINNED]: layout.aa_code_zero_based), k-mer = the last k residues, first one most significant,
// mask_aa_bits (utils.rs:66-76), add_kmer as for nucleotides (utils.rs:395-434).  Proteins are short and many: a LANE walks one
// record with a rolling register, a workgroup takes a range of a genome's records and shares one sketch in LDS.
// ------------------------------------------------------------------------------------------------------------
// Round 4.  The first version gave every lane a stride of records and walked each record in a loop of its own: the lanes of a wave
// then run in lockstep per RECORD, and a wave is as slow as its longest protein (50..2000 residues: 52 % of the lanes busy on
// average, tools/aa_rate.py), with a tail where the lanes' totals differ.  Now a lane is a little state machine in ONE loop — 16
// bytes of its record per trip, or the fetch of its next record — and records are handed out by a counter in LDS (one returning
// atomic per record), so every lane is busy until the item runs out.  Residue codes come from a 256-byte table in LDS (upper-casing,
// the 20-letter filter and the code in one ds_read_u8), deleted bytes and the bytes past a record's end simply do not advance the
// window (no branch), and the register rules run in their 32-bit fast forms with the exact form for the lanes that ask for it.
template <int ALGO, bool XLOW, int REGS>
__global__ void __launch_bounds__(1024) aa_sketch_kernel(SketchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_regs[];
    const uint32_t item = blockIdx.x + a.item_base;
    const WorkItem it = a.items[item];
    const GenomeDesc gd = a.genomes[it.genome];
    const int k = a.k, p = a.p;
    constexpr bool USE_LDS = REGS != REGS_GLOBAL;
    using Regs = typename std::conditional<REGS == REGS_LDS, LdsRegs,
                                           typename std::conditional<REGS == REGS_GLOBAL, GlobalRegs,
                                           typename std::conditional<REGS == REGS_BINS, BinRegs, LdsByteRegs>::type>::type>::type;
    Regs regs;
    uint32_t *census;
    const uint32_t part = 0u;
    if constexpr (REGS == REGS_BINS) {
        regs = bin_regs_of<ALGO>(a, it.genome);
        census = lds_regs;
    } else if constexpr (REGS == REGS_BYTES) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_regs != 0u) __builtin_trap();
        regs.p = p;
        census = lds_regs + a.nreg32;
        for (uint32_t i = threadIdx.x; i < a.nreg32; i += blockDim.x) lds_regs[i] = ALGO == 2 ? 0u : RANK_EMPTY;
    } else if constexpr (USE_LDS) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_regs != 0u) __builtin_trap();
        regs.base = lds_regs;
        census = lds_regs + a.nreg32;
        for (uint32_t i = threadIdx.x; i < a.nreg32; i += blockDim.x) lds_regs[i] = ALGO == 2 ? 0u : RANK_EMPTY;
    } else {
        regs.base = a.gregs + (uint64_t)(item - a.item_base) * a.nreg32;   // (tables of the items of this launch: lash_api.hip, global_run)
        census = lds_regs;
    }
    // after the census + histogram words (16 + 72): the record counter, then the byte -> code table
    uint32_t *const next_rec = census + 88;
    uint8_t *const code_tab = reinterpret_cast<uint8_t *>(census + 92);
    constexpr uint32_t LETTERS = 0x016FBDFDu;                                  // bit i: 'A' + i is one of ACDEFGHIKLMNPQRSTVWY
    const uint32_t base = a.lay.aa_code_base;
    if (threadIdx.x < 256u) {
        uint32_t c = threadIdx.x;
        c &= ~(((c - 0x61u) < 26u) ? 0x20u : 0u);                              // to_ascii_uppercase (utils.rs:43-55)
        const uint32_t idx = c - 0x41u;
        const bool letter = idx < 26u && ((LETTERS >> idx) & 1u);              // filter_out_a
        code_tab[threadIdx.x] = letter ? (uint8_t)((uint32_t)__builtin_popcount(LETTERS & ((1u << idx) - 1u)) + base) : (uint8_t)0xFFu;
    }
    if (threadIdx.x == 0) *next_rec = 0u;
    __syncthreads();
    const uint64_t kmask = (1ull << (5 * k)) - 1ull;                          // mask_aa_bits, k <= 12
    const uint32_t kmask_lo = (uint32_t)kmask, kmask_hi = (uint32_t)(kmask >> 32);
    const BitFlip bitflip = BitFlip::vector(a.bitflip);
    const uint32_t n_rec = it.word_end - it.word_begin;
    const uint64_t rec0 = gd.rec_begin + it.word_begin, genome_end = gd.byte_off + gd.byte_len;
    uint32_t my_kmers = 0;
    uint64_t i = 0, b1 = 0;                                                    // this lane's position in its record, the record's end
    uint32_t v_lo = 0, v_hi = 0, have = 0;
    bool done = false;
    for (;;) {
        if (!done && i >= b1) {                                                // fetch the next record
            const uint32_t r = __hip_atomic_fetch_add(next_rec, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (r >= n_rec) done = true;
            else {
                i = a.rec_off[rec0 + r]; b1 = a.rec_off[rec0 + r + 1];
                if (b1 - i < (uint64_t)k) i = b1;                              // utils.rs:523-525: the RAW length decides
                v_lo = v_hi = have = 0;
            }
        }
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        // (a lane between records, or out of them, runs the 16 steps below with nothing valid: no divergent skip — the binned
        // register mode ends every trip with a wave-wide flush)
        const bool act = !done && i < b1;
        const uint32_t n = !act ? 0u : (b1 - i >= 16 ? 16u : (uint32_t)(b1 - i));
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (act) {
            if (i + 16 <= genome_end) {                                        // (may run into the next record: those bytes are masked)
                const uint4 q = load16_any(a.seq + i);
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            } else {                                                           // the genome's last bytes: never past the caller's buffer
                for (uint32_t j = 0; j < n; ++j) w[j >> 2] |= (uint32_t)a.seq[i + j] << (8 * (j & 3));
            }
        }
        i += n;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t code = code_tab[(w[j >> 2] >> (8 * (j & 3))) & 0xFFu];
            const bool valid = code != 0xFFu && (uint32_t)j < n;               // a deleted byte joins its flanks: the window just does not move
            if constexpr (ALGO == 0) {
                v_lo = valid ? (v_lo << 5) | code : v_lo;                      // HyperMinHash hashes (masked as u32): the low word is all it needs
            } else {
                const uint32_t nh = alignbit(v_hi, v_lo, 27), nl = (v_lo << 5) | code;
                v_hi = valid ? nh : v_hi; v_lo = valid ? nl : v_lo;
            }
            have += valid ? 1u : 0u;
            const bool emit = valid && have >= (uint32_t)k;
            const uint32_t vm = emit ? 0xFFFFFFFFu : 0u;
            const uint32_t m_lo = v_lo & kmask_lo, m_hi = ALGO == 0 ? 0u : (v_hi & kmask_hi);
            const uint32_t t = add_kmer<ALGO, XLOW, true, true>(regs, m_lo, m_hi, vm, bitflip, p);
            constexpr uint32_t Z_REDO = z_redo<ALGO, Regs>();
            if (t <= Z_REDO) (void)add_kmer<ALGO, XLOW, true, false>(regs, m_lo, m_hi, vm, bitflip, p);   // (rare: the exact form)
            my_kmers += emit ? 1u : 0u;
        }
        if constexpr (REGS == REGS_BINS) regs.flush(threadIdx.x & 63u);
    }
    finish_item<ALGO, REGS, Regs>(a, it, regs, census, part, wave_sum(my_kmers), p, item);
}

template <int ALGO, bool XLOW>
static hipError_t launch_aa_regs(const SketchPlan &plan, const SketchArgs &args, uint32_t n, hipStream_t s)
{
    auto go = [&](auto kern) {
        SketchArgs a = args;
        a.bin_lds_off = plan.lds_bytes + 16u + 256u;                             // + the record counter and the byte -> code table
        a.bin_wave_bytes = sketch_bin_wave_bytes(plan);
        const uint32_t lds = a.bin_lds_off + (plan.threads / 64u) * a.bin_wave_bytes;
        if (lds > 48u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(n), dim3(plan.threads), lds, s, a);
        return hipGetLastError();
    };
    if (plan.bytes) return go(aa_sketch_kernel<ALGO, XLOW, REGS_BYTES>);
    if (plan.bins) return go(aa_sketch_kernel<ALGO, XLOW, REGS_BINS>);
    if (plan.use_lds) return go(aa_sketch_kernel<ALGO, XLOW, REGS_LDS>);
    return go(aa_sketch_kernel<ALGO, XLOW, REGS_GLOBAL>);
}

hipError_t launch_sketch_aa(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream)
{
    if (n_items == 0) return hipSuccess;
    switch (plan.algo) {
    case 0: return plan.variant ? launch_aa_regs<0, true>(plan, args, n_items, stream) : launch_aa_regs<0, false>(plan, args, n_items, stream);
    case 1: return plan.variant ? launch_aa_regs<1, true>(plan, args, n_items, stream) : launch_aa_regs<1, false>(plan, args, n_items, stream);
    case 2: return launch_aa_regs<2, false>(plan, args, n_items, stream);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------------------
// finalize: reduce a genome's partial sketches (max / ULL merge), add the image header, optionally union into
// what is already in the image (LASH_F_ACCUMULATE, lash_merge_images).  One workgroup per genome.
// ------------------------------------------------------------------------------------------------------------
// One level of the fold: units are slices (stride 1) or the heads an earlier level left (stride = slices per head); every
// R consecutive units are merged into the first one's partial.
template <int ALGO>
__global__ void __launch_bounds__(256) reduce_groups_kernel(FinalizeArgs a, uint32_t R, uint32_t stride)
{
    const uint32_t g = blockIdx.y, P = 1u << a.parts_log2, part = blockIdx.x % P, j = blockIdx.x / P;
    const uint32_t i0 = a.genome_item_begin[g], i1 = a.genome_item_begin[g + 1];
    const uint32_t n_slices = (i1 - i0) / P, n_units = (n_slices + stride - 1) / stride, u0 = j * R;
    if (u0 >= n_units) return;
    const uint32_t u1 = u0 + R < n_units ? u0 + R : n_units;
    uint64_t nk = ~0ull;
    if (a.nvalid) { const uint64_t L = a.nvalid[g]; nk = L >= (uint64_t)a.k ? L - (uint64_t)a.k + 1 : 0; }
    auto live = [&](uint32_t it) { return stride > 1u || (uint64_t)a.items[it].word_begin * 16 < nk; };   // heads always are
    const uint32_t nwords = (ALGO == 0 ? HMH_M * 2 : (1u << a.p)) >> 2, nw_part = nwords >> a.parts_log2, wbase = part * nw_part;
    const uint32_t head = i0 + u0 * stride * P + part;
    uint8_t *dst = const_cast<uint8_t *>(a.partials) + (uint64_t)head * a.partial_stride + a.partial_base_off;
    // blockIdx.z splits the pass's words so that a thread folds one word: R independent loads, no serial walk
    for (uint32_t wi = wbase + blockIdx.z * blockDim.x + threadIdx.x; wi < wbase + nw_part; wi += gridDim.z * blockDim.x) {
        uint32_t acc = 0;
#pragma unroll 8
        for (uint32_t u = u0; u < u1; ++u) {
            const uint32_t it = i0 + u * stride * P + part;
            if (!live(it)) continue;                                       // slice never ran: its partial is not defined
            acc = merge_word<ALGO>(acc, load_u32_any(a.partials + (uint64_t)it * a.partial_stride + a.partial_base_off + 4ull * wi));
        }
        store_u32_any(dst + 4ull * wi, acc);                               // the head's own word was read above, by this thread
    }
    if (a.item_kmers && threadIdx.x == 0 && blockIdx.z == 0) {
        uint32_t tot = 0;
        for (uint32_t u = u0; u < u1; ++u) {
            const uint32_t it = i0 + u * stride * P + part;
            if (live(it)) tot += a.item_kmers[it];
        }
        const_cast<uint32_t *>(a.item_kmers)[head] = tot;
    }
}

// args.group = slices per head after all levels (a power of R): R = 32 slices per level until <= 64 heads remain
hipError_t launch_reduce_groups(const FinalizeArgs &args, uint32_t n_genomes, uint32_t max_slices, hipStream_t stream)
{
    if (n_genomes == 0 || args.group == 0) return hipSuccess;
    if (n_genomes > 65535u) return hipErrorInvalidValue;                   // (the caller only groups when genomes are few)
    constexpr uint32_t R = 32;
    for (uint32_t stride = 1; stride < args.group; stride *= R) {
        const uint32_t units = (max_slices + stride - 1) / stride, groups = (units + R - 1) / R;
        const uint32_t nw_part = ((args.algo == 0 ? HMH_M * 2 : (1u << args.p)) >> 2) >> args.parts_log2;
        dim3 grid(groups << args.parts_log2, n_genomes, std::max(1u, std::min(nw_part / 256u, 64u)));
        switch (args.algo) {
        case 0: hipLaunchKernelGGL(reduce_groups_kernel<0>, grid, dim3(256), 0, stream, args, R, stride); break;
        case 1: hipLaunchKernelGGL(reduce_groups_kernel<1>, grid, dim3(256), 0, stream, args, R, stride); break;
        case 2: hipLaunchKernelGGL(reduce_groups_kernel<2>, grid, dim3(256), 0, stream, args, R, stride); break;
        default: return hipErrorInvalidValue;
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int ALGO>
__global__ void __launch_bounds__(1024) finalize_kernel(FinalizeArgs a)
{
    __shared__ uint32_t hist[72];
    const uint32_t g = blockIdx.x;
    if (a.descs && a.descs[g].byte_len <= a.skip_max_len) return;         // the persistent small-genome kernel's (sole_kernels.hip)
    const uint32_t i0 = a.genome_item_begin[g], i1 = a.genome_item_begin[g + 1];
    uint64_t nk = ~0ull;
    if (a.nvalid) { const uint64_t L = a.nvalid[g]; nk = L >= (uint64_t)a.k ? L - (uint64_t)a.k + 1 : 0; }
    const uint32_t hdr = a.lay.hdr_bytes, reg_be = ALGO == 0 ? a.lay.hmh_reg_be : 0u;
    const uint32_t src_be = a.src_images ? reg_be : 0u;                    // merge of images: the "partials" are in image byte order
    const uint32_t nbytes = ALGO == 0 ? HMH_M * 2 : (1u << a.p);
    const uint32_t nwords = nbytes >> 2;                                   // p >= 3 -> at least 2 words
    uint8_t *img = a.images + (uint64_t)g * a.image_bytes;
    if (threadIdx.x < 72) hist[threadIdx.x] = 0;
    __syncthreads();
    // items of a genome are ordered slice-major, pass-minor: item = i0 + slice * P + pass.  After launch_reduce_groups()
    // only every `group`-th slice (a group head, always treated as live) still matters.
    const uint32_t P = 1u << a.parts_log2, step = (a.group ? a.group : 1u) * P;
    if (a.item_kmers && threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (uint32_t it = i0; it < i1; it += step)                        // pass 0 carries the count of its slice
            if (a.group || (uint64_t)a.items[it].word_begin * 16 < nk) tot += a.item_kmers[it];
        if (tot) atomicAdd(a.kmer_counter, tot);
    }
    // a genome sketched by a single work item has had its image written by that item (ITEM_SOLE, sketch_kernel)
    if (i1 - i0 == 1u && (a.items[i0].slice & ITEM_SOLE)) return;

    HllTally tally;
    for (uint32_t wi = threadIdx.x; wi < nwords; wi += blockDim.x) {
        uint32_t acc = a.accumulate ? hmh_img_order(load_u32_any(img + hdr + 4ull * wi), reg_be) : 0u;
        const uint32_t part = a.parts_log2 ? wi / (nwords >> a.parts_log2) : 0u;   // whose pass wrote this word
        for (uint32_t it = i0 + part; it < i1; it += step) {
            if (!a.group && (uint64_t)a.items[it].word_begin * 16 >= nk) continue;   // slice never ran (see sketch_kernel)
            const uint8_t *src = a.partials + (uint64_t)it * a.partial_stride + a.partial_base_off;
            acc = merge_word<ALGO>(acc, hmh_img_order(load_u32_any(src + 4ull * wi), src_be));
        }
        store_u32_any(img + hdr + 4ull * wi, hmh_img_order(acc, reg_be));
        if constexpr (ALGO == 1) tally.add(hist, acc);
    }
    if constexpr (ALGO == 1) {
        tally.flush(hist);
        __syncthreads();
        if (threadIdx.x < 64u) write_hll_header_wave(img, a.lay.hdr_tpl, hist, a.alpha_bits, a.p, a.hll_corner ? a.hll_corner + g : nullptr);
    } else {
        if (threadIdx.x == 0 && hdr) write_header(img, a.lay.hdr_tpl, a.alpha_bits, ALGO == 0 ? HMH_M : (1ull << a.p), 0, 0.0, ALGO == 0 ? HMH_P : a.p);
    }
}

// ------------------------------------------------------------------------------------------------------------
// bins_apply_kernel — one workgroup per (genome, bin) of a binned launch: the bin's registers are built in LDS from the bin's
// list (six 21-bit entries per 16-byte chunk, read once, coalesced), the genome's fallback table is read beside it if anything was spilled
// there, and the registers leave as image-format bytes — UltraLogLog, not accumulating: straight into the caller's image (header and k-mer
// census too, no finalize launch); otherwise into the genome's one partial sketch, and finalize_kernel does the rest as for any genome.
// ------------------------------------------------------------------------------------------------------------
template <int ALGO>
__global__ void __launch_bounds__(1024) bins_apply_kernel(BinApplyArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];
    const uint32_t bin = blockIdx.x, gi = blockIdx.y;
    // ONE 32-bit word per register.  HyperLogLog: rho - 1 under max.  UltraLogLog: the nlz bitmap's low word — an entry with 32 or more leading
    // zeros (2^-32 of the k-mers) waits in a short list beside the table and is merged when the registers are written out
    // (round 6: with 64-bit bitmaps a bin was 2^14 registers; at a word each it is 2^15 — half as many bins for the sketch kernels to scatter over)
    const uint32_t regs_per_bin = 1u << a.bin_shift, words = regs_per_bin, slab_words_bin = ALGO == 2 ? 2u * regs_per_bin : regs_per_bin;
    uint32_t *const rare = tab + words;                                   // [0]: how many, [1 ..]: the entries (BINS_APPLY_RARE of them)
    {
        const uint32_t fill = ALGO == 2 ? 0u : RANK_EMPTY;
        uint4 *t4 = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = threadIdx.x; i < (words >> 2); i += blockDim.x) t4[i] = make_uint4(fill, fill, fill, fill);
        if (threadIdx.x == 0u) rare[0] = 0u;
    }
    __syncthreads();
    const BinGenome bg = a.genomes[gi];
    const uint32_t *list = a.lists + bg.list_off + (uint64_t)bin * bg.cap;
    uint32_t n = a.cnt[(uint64_t)gi * a.bins + bin];
    n = n < bg.cap ? n : bg.cap;
    uint32_t *const sl = a.slab + (uint64_t)gi * a.slab_words + (uint64_t)bin * slab_words_bin;
    uint32_t *const flag = a.spill + (uint64_t)gi * a.bins + bin;
    // 16 x LASH_BINS_APPLY_LOADS bytes per lane and round, all loads issued before the first update, no branch ("nothing" ORs a zero / offers
    // -1): one entry per round with a `continue` in it ran one global load latency per entry — 76 in a row per lane at p = 20,
    // 129 us per workgroup, half of a binned launch's time
    // a list entry (round 6): index inside the bin << 6 | value in bin_shift + 6 bits, six of them in a 16-byte chunk (BinRegs::flush)
    uint32_t rare_seen = 0;
    auto apply = [&](uint32_t f) {
        // ("nothing" — a row's padding, a round's idle lanes — goes to a register of the lane's own: the lanes' zeros on ONE word serialise)
        const uint32_t v = f & 63u, r = (v == 63u ? threadIdx.x : (f >> 6)) & (regs_per_bin - 1u);
        if constexpr (ALGO == 2) {
            atomicOr(&tab[r], v >= 32u ? 0u : 1u << v);
            rare_seen |= (v + 1u) & 63u;                                   // 33 .. 63 for nlz = 32 .. 62, at most 32 otherwise
        } else {
            atomicMax(reinterpret_cast<int *>(tab) + r, v == 63u ? -1 : (int)v);
        }
    };
    // the rare entry: into the short list; a list that is full (never, with hashed input) sends it to the genome's fallback table instead
    auto apply_rare = [&](uint32_t f) {
        const uint32_t v = f & 63u, r = (f >> 6) & (regs_per_bin - 1u);
        if (((v + 1u) & 63u) <= 32u) return;
        const uint32_t slot = atomicAdd(&rare[0], 1u);
        if (slot < BINS_APPLY_RARE) rare[1u + slot] = (r << 6) | v;
        else {
            (void)__hip_atomic_fetch_or(sl + 2u * r + 1u, 1u << (v - 32u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const uint32_t F = a.bin_shift + 6u, fmask = (1u << F) - 1u;
    auto for_fields = [&](const uint4 q, auto fn) {
        const uint64_t lo = ((uint64_t)q.y << 32) | q.x, hi = ((uint64_t)q.w << 32) | q.z;
        fn((uint32_t)lo & fmask); fn((uint32_t)(lo >> F) & fmask); fn((uint32_t)(lo >> (2u * F)) & fmask);
        fn((uint32_t)hi & fmask); fn((uint32_t)(hi >> F) & fmask); fn((uint32_t)(hi >> (2u * F)) & fmask);
    };
    const uint4 *l4 = reinterpret_cast<const uint4 *>(list);          // (lists start and reservations are whole 16-byte chunks)
    const uint32_t n4 = n >> 2;
    constexpr uint32_t Q = LASH_BINS_APPLY_LOADS;
    for (uint32_t i = threadIdx.x; i < n4; i += Q * blockDim.x) {
        uint4 q[Q];
#pragma unroll
        for (uint32_t b = 0; b < Q; ++b) {
            const uint32_t at = i + b * blockDim.x;
            q[b] = at < n4 ? l4[at] : make_uint4(~0u, ~0u, ~0u, ~0u);      // (all ones: six times "nothing")
        }
        // (a wave whose 64 chunks all lie beyond the list skips them: the last round of a list is on average half empty, and its "nothing"
        // entries are LDS atomics like any other — Q = 8 lost 10 % to them, Q = 2 won 3 %, profiles/r06/bins_ab.txt)
        const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i - (threadIdx.x & 63u)));
#pragma unroll
        for (uint32_t b = 0; b < Q; ++b)
            if (wave_first + b * blockDim.x < n4) for_fields(q[b], apply);
        if constexpr (ALGO == 2) {
            if (__builtin_expect(rare_seen > 32u, 0)) {
#pragma unroll
                for (uint32_t b = 0; b < Q; ++b) for_fields(q[b], apply_rare);
            }
            rare_seen = 0;
        }
    }
    __syncthreads();
    uint32_t n_rare = 0;
    if constexpr (ALGO == 2) {
        n_rare = rare[0];
        if (__builtin_expect(n_rare > BINS_APPLY_RARE, 0)) {              // (workgroup-uniform) this workgroup's own entries in the fallback table: an
            __threadfence();                                              // agent-scope release — a whole L2 write-back on this chip, hence never
            __syncthreads();                                              // on the common path
            n_rare = BINS_APPLY_RARE;
        }
    }
    // entries of this bin that found a row or a list full sit in this bin's part of the genome's fallback table: read beside the LDS table and
    // left empty again — the table is wiped once, when it is allocated, not 8 MiB per genome and call (p = 20; with ONE flag per genome
    // every bin of nearly every genome read its part: some row overflows somewhere)
    const bool spilled = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    auto slab_take = [&](uint32_t i, uint32_t empty) {                    // (agent scope: past this CU's vector cache)
        const uint32_t x = __hip_atomic_load(sl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x != empty) sl[i] = empty;
        return x;
    };
    uint32_t *out = reinterpret_cast<uint32_t *>(a.partials + (uint64_t)(a.genome0 + gi) * a.partial_stride) + (uint64_t)bin * (regs_per_bin >> 2);
    uint8_t *const img = a.images ? a.images + (uint64_t)(a.genome0 + gi) * a.image_bytes : nullptr;
    uint8_t *const img_out = img ? img + a.hdr_bytes + (uint64_t)bin * regs_per_bin : nullptr;
    for (uint32_t i = threadIdx.x; i < (regs_per_bin >> 2); i += blockDim.x) {
        const uint4 t = reinterpret_cast<const uint4 *>(tab)[i];
        const uint32_t tv[4] = {t.x, t.y, t.z, t.w};
        uint32_t o = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
            uint32_t r;
            if constexpr (ALGO == 2) {
                uint32_t lo = tv[b], hi = 0u;
                if (spilled) { lo |= slab_take(8u * i + 2u * b, 0u); hi = slab_take(8u * i + 2u * b + 1u, 0u); }
                for (uint32_t j = 0; j < n_rare; ++j) {                    // (n_rare is zero but once in 2^32 k-mers)
                    const uint32_t e = rare[1u + j];
                    if ((e >> 6) == 4u * i + b) hi |= 1u << ((e & 63u) - 32u);
                }
                r = 0;
                if (lo | hi) {                                             // as finish_item: nlz bitmap -> hash4j prefix -> pack()
                    const uint64_t x = (((uint64_t)hi << 32) | lo) << (a.p - 1);
                    const uint32_t top = 63u - (uint32_t)__builtin_clzll(x);
                    const uint32_t below = top >= 2 ? (uint32_t)(x >> (top - 2)) & 3u : (uint32_t)(x << (2 - top)) & 3u;
                    r = (top << 2) | below;
                }
            } else {
                int m = (int)tv[b];
                if (spilled) { const int s2 = (int)slab_take(4u * i + b, RANK_EMPTY); m = m > s2 ? m : s2; }
                r = (uint32_t)m + 1u;                                      // rho - 1, -1 = empty
            }
            o |= (r & 0xFFu) << (8 * b);
        }
        if (img_out) store_u32_any(img_out + 4ull * i, o);
        else out[i] = o;
    }
    if (bin == 0u && threadIdx.x == 0u) {                                  // the genome's k-mer census: the sum over its work items
        const uint32_t g = a.genome0 + gi;
        uint64_t nk = ~0ull;
        if (a.nvalid) { const uint64_t L = a.nvalid[g]; nk = L >= (uint64_t)a.k ? L - (uint64_t)a.k + 1 : 0; }
        unsigned long long tot = 0;                                        // (a genome beyond 2^32 k-mers: only this statistic would notice)
        for (uint32_t it = a.genome_item_begin[g]; it < a.genome_item_begin[g + 1]; ++it)
            if ((uint64_t)a.items[it].word_begin * 16 < nk) tot += a.item_kmers[it];      // (a slice beyond the surviving bases never ran)
        a.item_kmers[a.virt0 + gi] = (uint32_t)tot;
        if (img) {                                                         // (what finalize_kernel would have done for this genome)
            if (tot) atomicAdd(a.kmer_counter, tot);
            if (a.hdr_bytes) write_header(img, a.hdr_tpl, 0ull, 1ull << a.p, 0, 0.0, a.p);
        }
    }
}

hipError_t launch_bins_apply(const BinApplyArgs &args, uint32_t n_group_genomes, hipStream_t stream)
{
    if (n_group_genomes == 0) return hipSuccess;
    const uint32_t words = 1u << args.bin_shift, lds = (words + 1u + BINS_APPLY_RARE) * 4u;   // one word per register + the short list of rare entries
    auto go = [&](auto kern) {
        if (lds > 48u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(args.bins, n_group_genomes), dim3(1024), lds, stream, args);
        return hipGetLastError();
    };
    return args.algo == 2 ? go(bins_apply_kernel<2>) : go(bins_apply_kernel<1>);
}

// ------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------------------
SketchPlan make_sketch_plan(int algo, int k, int p, bool variant, bool small_items, bool allow_bins)
{
    SketchPlan s{};
    s.algo = algo; s.k = k; s.p = p; s.variant = variant && algo != 2;
    if (algo == 0) { s.nreg32 = HMH_M; s.partial_bytes = HMH_M * 2; }
    else if (algo == 1) { s.nreg32 = 1u << p; s.partial_bytes = 1u << p; }
    else { s.nreg32 = 2u << p; s.partial_bytes = 1u << p; }
    s.partial_stride = (s.partial_bytes + 15u) & ~15u;
    s.lds_bytes = s.nreg32 * 4u;
    s.parts_log2 = 0;
    // tables beyond 128 KiB: binned (BinRegs; up to 256 bins: HLL p = 16, ULL p = 15 .. 23), beyond that a table in global
    // memory per work item and one atomic per k-mer (ULL p >= 24)
    uint32_t bl = 0;
    while ((s.lds_bytes >> bl) > 128u * 1024u) ++bl;
    const bool no_bins = getenv("LASH_NO_BINS") != nullptr;                // A/B knob (read per call): the global-atomic path for every large table
    const bool no_bytes = getenv("LASH_NO_BYTES") != nullptr;              // A/B knob (read per call): bins instead of byte tables
    // up to 128 KiB of BYTE registers (hll p = 16, ull p = 15 .. 17): one pass, compare-and-swap updates (LdsByteRegs); binning pays from 8 bins on
    s.bytes = bl > 0 && (1u << p) <= 128u * 1024u && algo != 0 && !no_bytes;
    if (s.bytes) { bl = 0; s.nreg32 = (1u << p) >> 2; s.lds_bytes = s.nreg32 * 4u; }
    // bins of 2^15 registers: bins_apply_kernel holds a 32-bit word per register, 128 KiB of LDS (round 6: UltraLogLog too — its bitmap's high word is
    // the rare entries' short list —, so half as many bins as with 64-bit words: fuller staging rows per flush, fuller chunks; and p = 23 fits 256 bins)
    uint32_t bin_shift = 15u;
    if (algo == 2 && bl > 0) {
        if (bl > 1u) --bl; else bin_shift = 14u;
        if (const char *e = getenv("LASH_BIN_SHIFT")) {                     // A/B knob (read per call): 14 = bins of 2^14 registers as in round 5
            if (atoi(e) == 14 && bin_shift == 15u) { bin_shift = 14u; ++bl; }
        }
    }
    s.bins = bl > 0 && bl <= 8u && !no_bins && allow_bins;
    s.use_lds = bl == 0 || s.bins;
    if (s.bins) {
        s.bin_shift = bin_shift;
        s.bins_log2 = bl;
        s.bin_sub_shift = bl < 5u ? 5u - bl : 0u;                           // at least 32 staging rows per wave (see BinRegs; 64: slower at p = 18 .. 20, profiles/r06/bins_ab.txt)
        // 64 bins (p = 21) and 256 (p = 23): a row collects TWO words' entries before it leaves — fuller chunks, half as many list reservations and
        // partial cache lines: -5 % and -19 % (profiles/r06/bins_ab.txt section 10).  Not at 128 bins: the rows of p = 22 would then take a CU's LDS with
        // eight waves instead of twelve (+12 %); not below 64 bins: rows are full enough (+9 % at p = 20).  (A/B knob LASH_BIN_FLUSH = 1 / 2 / 4)
        s.bin_flush_words = (bl == 6u || bl == 8u) ? 2u : 1u;
        if (const char *e = getenv("LASH_BIN_FLUSH")) { const int f = atoi(e); if (f == 1 || f == 2 || f == 4) s.bin_flush_words = (uint32_t)f; }
        for (;; s.bin_flush_words >>= 1) {
            const uint32_t mean = (1024u * s.bin_flush_words) >> (bl + s.bin_sub_shift);   // staged entries per row between two flushes (a word = 16 k-mers per lane)
            uint32_t sq = 1; while (sq * sq < mean) ++sq;
            s.bin_S = ((mean + 4u * sq + 4u + 5u) / 6u) * 6u;              // room for the mean + 4 sigma, a multiple of six (six entries per chunk); more goes to the fallback table
            const uint32_t wave_bytes = ((1u << (bl + s.bin_sub_shift)) * (1u + bin_row_stride(s.bin_S))) * 4u;
            if (s.bin_flush_words == 1u || wave_bytes <= (bl >= 6u ? 39u : 19u) * 1024u) break;   // (a workgroup's four / eight waves of rows must fit a CU's LDS)
        }
        s.lds_bytes = 0;                                                    // no table in the sketch kernels
    }
    s.threads = (s.use_lds && s.lds_bytes > 64u * 1024u) ? 1024u : 512u;  // <=64 KiB: two workgroups per CU
    // 64 bins and more: a wave's staging rows take 9 .. 17 KiB — workgroups of four waves, so that two or three of them share a CU
    if (s.bins && bl >= 6u) s.threads = 256u;
    // Small genomes with a small table (hll p<=13, ull p<=12): a 10 kbp genome fills 2.5 waves, and what limits such a
    // batch is per-workgroup latency (item / descriptor / first tile loads, flush), not issue slots -> 256-thread
    // workgroups, as many per CU as the table allows (100 000 x 10 kbp, hll p=10: 4.4 -> 2.7 ms).  HyperMinHash's 64 KiB
    // table admits two workgroups per CU whatever their size, and smaller ones only lose lanes.
    const bool many_small = small_items && s.use_lds && s.parts_log2 == 0 && s.lds_bytes <= 32u * 1024u;
    if (many_small) s.threads = 256u;
    if (const char *e = getenv("LASH_SKETCH_THREADS")) {                    // tuning knob (tools/)
        const int t = atoi(e);
        if (t >= 64 && t <= 1024 && t % 64 == 0) s.threads = (uint32_t)t;
    }
    if (const char *e = getenv("LASH_SIGQ_DEPTH")) {                        // tuning knob: words of a lane's stack in deferring launches
        const int d = atoi(e) | 1;
        if (d >= (int)SIGQ_MIN_DEPTH && d <= 63) s.sigq_depth = (uint32_t)d;
    }
    if (!s.use_lds) s.lds_bytes = 0;
    else if (s.bins) {}
    // Small tables (hll p<=13, ull p<=12) would let 4 workgroups = 8 waves/SIMD share a CU; the kernel is VALU-issue
    // bound and the extra waves only add LDS-atomic contention (hll p=13: 7.4e11 vs 8.1e11 k-mers/s).  Asking for 64 KiB
    // keeps it at the two workgroups per CU that the 64 KiB tables get.
    else if (!many_small && !s.bins && s.threads == 512u && s.lds_bytes < 64u * 1024u) s.lds_bytes = 64u * 1024u;
    s.lds_bytes += 64u + 288u;                                             // per-wave census words + HLL header histogram after the registers
    return s;
}

// per wave: dense_tile's staging area (direct mode) and / or the lanes' stacks of a deferring launch (they share it)
static uint32_t stage_stride_bytes(const SketchPlan &plan, bool direct, bool defer)
{
    const uint32_t stage = direct ? DENSE_STAGE_WORDS * 4u : 0u, stacks = defer ? 64u * plan.sigq_depth * 4u : 0u;
    return std::max(stage, stacks);
}
uint32_t sketch_direct_stage_bytes(const SketchPlan &plan) { return (plan.threads / 64u) * stage_stride_bytes(plan, true, true); }
uint32_t sketch_bin_wave_bytes(const SketchPlan &plan) { return plan.bins ? ((1u << (plan.bins_log2 + plan.bin_sub_shift)) * (1u + bin_row_stride(plan.bin_S))) * 4u : 0u; }

template <int ALGO, int KMODE, bool XLOW, int REGS, bool DIRECT>
static hipError_t launch_one(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream)
{
    auto kern = sketch_kernel<ALGO, KMODE, XLOW, REGS, DIRECT>;
    bool defer = false;
    if constexpr (ALGO == 0 && REGS == REGS_LDS) {
        if (plan.defer) { kern = sketch_kernel<ALGO, KMODE, XLOW, REGS, DIRECT, true>; defer = true; }
    }
    SketchArgs a = args;
    a.stage_off = plan.lds_bytes;                                          // direct mode: the waves' staging areas follow (dense_tile);
    a.stage_stride = stage_stride_bytes(plan, DIRECT, defer || REGS == REGS_BYTES);   // deferring launches keep their lanes' stacks there, the byte tables' theirs
    a.sigq_depth = plan.sigq_depth;
    a.bin_lds_off = plan.lds_bytes + (plan.threads / 64u) * a.stage_stride;
    a.bin_wave_bytes = sketch_bin_wave_bytes(plan);
    const uint32_t lds = a.bin_lds_off + (plan.threads / 64u) * a.bin_wave_bytes;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_items), dim3(plan.threads), lds, stream, a);
    return hipGetLastError();
}

template <int ALGO, bool XLOW, bool DIRECT>
static hipError_t launch_kmode(const SketchPlan &plan, const SketchArgs &args, uint32_t n, hipStream_t s)
{
    const int km = plan.k == 16 ? KM_16 : plan.k < 16 ? KM_LT16 : KM_GT16;
    if constexpr (ALGO != 0) {                                  // HMH's table always fits
        if (plan.bytes) {
            if (km == KM_16) return launch_one<ALGO, KM_16, XLOW, REGS_BYTES, DIRECT>(plan, args, n, s);
            if (km == KM_LT16) return launch_one<ALGO, KM_LT16, XLOW, REGS_BYTES, DIRECT>(plan, args, n, s);
            return launch_one<ALGO, KM_GT16, XLOW, REGS_BYTES, DIRECT>(plan, args, n, s);
        }
        if (plan.bins) {
            if (km == KM_16) return launch_one<ALGO, KM_16, XLOW, REGS_BINS, DIRECT>(plan, args, n, s);
            if (km == KM_LT16) return launch_one<ALGO, KM_LT16, XLOW, REGS_BINS, DIRECT>(plan, args, n, s);
            return launch_one<ALGO, KM_GT16, XLOW, REGS_BINS, DIRECT>(plan, args, n, s);
        }
    }
    if (plan.use_lds) {
        if (km == KM_16) return launch_one<ALGO, KM_16, XLOW, REGS_LDS, DIRECT>(plan, args, n, s);
        if (km == KM_LT16) return launch_one<ALGO, KM_LT16, XLOW, REGS_LDS, DIRECT>(plan, args, n, s);
        return launch_one<ALGO, KM_GT16, XLOW, REGS_LDS, DIRECT>(plan, args, n, s);
    }
    if constexpr (ALGO == 2) {                                  // only ULL p >= 19 outgrows the partitioned LDS passes
        if (km == KM_16) return launch_one<ALGO, KM_16, XLOW, REGS_GLOBAL, DIRECT>(plan, args, n, s);
        if (km == KM_LT16) return launch_one<ALGO, KM_LT16, XLOW, REGS_GLOBAL, DIRECT>(plan, args, n, s);
        return launch_one<ALGO, KM_GT16, XLOW, REGS_GLOBAL, DIRECT>(plan, args, n, s);
    }
    return hipErrorInvalidValue;
}

template <bool DIRECT>
static hipError_t launch_algo(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream)
{
    switch (plan.algo) {
    case 0: return plan.variant ? launch_kmode<0, true, DIRECT>(plan, args, n_items, stream) : launch_kmode<0, false, DIRECT>(plan, args, n_items, stream);
    case 1: return plan.variant ? launch_kmode<1, true, DIRECT>(plan, args, n_items, stream) : launch_kmode<1, false, DIRECT>(plan, args, n_items, stream);
    case 2: return launch_kmode<2, false, DIRECT>(plan, args, n_items, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sketch(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream, bool direct)
{
    if (n_items == 0) return hipSuccess;
    return direct ? launch_algo<true>(plan, args, n_items, stream) : launch_algo<false>(plan, args, n_items, stream);
}

// Are all records of a genome the same length (a FASTQ read set)?  Then record starts are the multiples of that length and the
// sketch kernel derives them arithmetically: no bitmap to make (brk_bytes_kernel: 6 % of a reads-shaped step), to clear or to read
// (0.125 B per base).  One thread per record compares its length with the genome's first record's.
__global__ void __launch_bounds__(256) rec_uniform_kernel(const GenomeDesc *genomes, const uint64_t *rec_off, uint32_t n_genomes,
                                                          uint32_t *nonuniform)
{
    for (uint32_t g = blockIdx.y; g < n_genomes; g += gridDim.y) {
        const GenomeDesc gd = genomes[g];
        if (gd.format != 0u || gd.rec_end - gd.rec_begin <= 1) continue;
        const uint64_t len0 = rec_off[gd.rec_begin + 1] - rec_off[gd.rec_begin];
        if (len0 == 0 || len0 > 0xFFFFFFFFull) { if (blockIdx.x == 0 && threadIdx.x == 0) nonuniform[g] = 1u; continue; }
        bool differs = false;
        for (uint64_t r = gd.rec_begin + 1 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < gd.rec_end; r += (uint64_t)gridDim.x * blockDim.x)
            differs = differs || rec_off[r + 1] - rec_off[r] != len0;
        if (differs) nonuniform[g] = 1u;                                 // (same value from every writer)
    }
}

// which of the two bitmap writers takes a genome: long records (a draft assembly: contigs of a KiB and more on average) are zeroed by
// every thread and have their few starts OR-ed in; short ones (a read set) are written word by word from their "head" records
__device__ __forceinline__ bool long_records(const GenomeDesc &gd) { return gd.byte_len / (gd.rec_end - gd.rec_begin) >= 1024u; }

// one thread per record: the first byte of every record but a genome's first is a k-mer barrier (utils.rs:457-464).
// Record offsets ascend, so the records whose bits share a 32-bit word are consecutive: the first of them ("head": the
// record before it lands in another word) gathers the bits of its followers, writes the word with ONE plain store and zeroes
// the words up to the next head's — every word of the genome's bitmap is written exactly once: no atomics (20 M atomicOr for a
// 3 Gbp read set took 0.29 ms, 8 % of the step), no memset.
__global__ void __launch_bounds__(256) brk_bytes_kernel(const GenomeDesc *genomes, const uint64_t *rec_off, uint32_t n_genomes,
                                                        const uint32_t *nonuniform, uint32_t *brk_bytes)
{
    for (uint32_t g = blockIdx.y; g < n_genomes; g += gridDim.y) {
        const GenomeDesc gd = genomes[g];
        if (gd.format != 0u || gd.rec_end - gd.rec_begin <= 1) continue;
        if (nonuniform && nonuniform[g] == 0u) continue;                  // equal-length records: nobody reads this genome's bitmap
        if (long_records(gd)) continue;                                   // brk_zero_kernel + brk_set_kernel
        const uint64_t n_words = (gd.byte_len + 1 + 31) / 32 + 4;        // (+3 words of look-ahead in kmer_valid_mask)
        uint32_t *bm = brk_bytes + gd.brk_off;
        // in-genome record starts: positions < byte_len (empty records at the genome's end start at byte_len: no barrier)
        auto pos_of = [&](uint64_t r) { return rec_off[r] - gd.byte_off; };
        for (uint64_t r = gd.rec_begin + 1 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < gd.rec_end;
             r += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t pos = pos_of(r);
            if (pos >= gd.byte_len) {                                     // nothing to mark; the first such record finishes the bitmap
                if (r == gd.rec_begin + 1 || pos_of(r - 1) < gd.byte_len) {
                    const uint64_t from = r == gd.rec_begin + 1 ? 0 : (pos_of(r - 1) >> 5) + 1;
                    for (uint64_t w = from; w < n_words; ++w) bm[w] = 0u;
                }
                continue;
            }
            const uint64_t word = pos >> 5;
            if (r > gd.rec_begin + 1 && (pos_of(r - 1) >> 5) == word) continue;      // a follower
            if (r == gd.rec_begin + 1) for (uint64_t w = 0; w < word; ++w) bm[w] = 0u; // before the first record start
            uint32_t bits = 1u << (pos & 31);
            uint64_t q = r + 1, next_word = n_words;
            for (; q < gd.rec_end; ++q) {
                const uint64_t pq = pos_of(q);
                if (pq >= gd.byte_len) break;
                if ((pq >> 5) != word) { next_word = pq >> 5; break; }
                bits |= 1u << (pq & 31);
            }
            bm[word] = bits;
            for (uint64_t w = word + 1; w < next_word; ++w) bm[w] = 0u;
        }
    }
}

// The same bitmap for genomes of FEW LONG records (a draft assembly: tens of contigs of 100 kb): the head-record form above would
// have each of 50 threads zero-fill 12 KiB on its own (3.3 ms per 1 000 genomes — more than sketching them); here every thread
// zeroes its share of the words and the few record starts are OR-ed in afterwards (0.2 ms).
__global__ void __launch_bounds__(256) brk_zero_kernel(const GenomeDesc *genomes, uint32_t n_genomes, const uint32_t *nonuniform, uint32_t *brk_bytes)
{
    for (uint32_t g = blockIdx.y; g < n_genomes; g += gridDim.y) {
        const GenomeDesc gd = genomes[g];
        if (gd.format != 0u || gd.rec_end - gd.rec_begin <= 1) continue;
        if (nonuniform && nonuniform[g] == 0u) continue;
        if (!long_records(gd)) continue;
        const uint64_t n_words = (gd.byte_len + 1 + 31) / 32 + 4;
        uint32_t *bm = brk_bytes + gd.brk_off;
        for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) bm[w] = 0u;
    }
}
__global__ void __launch_bounds__(256) brk_set_kernel(const GenomeDesc *genomes, const uint64_t *rec_off, uint32_t n_genomes,
                                                      const uint32_t *nonuniform, uint32_t *brk_bytes)
{
    for (uint32_t g = blockIdx.y; g < n_genomes; g += gridDim.y) {
        const GenomeDesc gd = genomes[g];
        if (gd.format != 0u || gd.rec_end - gd.rec_begin <= 1) continue;
        if (nonuniform && nonuniform[g] == 0u) continue;
        if (!long_records(gd)) continue;
        uint32_t *bm = brk_bytes + gd.brk_off;
        for (uint64_t r = gd.rec_begin + 1 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < gd.rec_end; r += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t pos = rec_off[r] - gd.byte_off;
            if (pos < gd.byte_len) atomicOr(bm + (pos >> 5), 1u << (pos & 31));
        }
    }
}

static dim3 per_record_grid(uint32_t n_genomes, uint64_t n_rec)
{
    // x: enough 256-thread blocks per genome for ~4 records per thread (a read set is one genome with millions of records)
    const uint64_t per_genome = n_rec / n_genomes + 1;
    const uint32_t gx = (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(8, per_genome / 1024 + 1));
    return dim3(gx, n_genomes < 65535u ? n_genomes : 65535u);
}

hipError_t launch_rec_uniform(const GenomeDesc *genomes, const uint64_t *rec_off, uint32_t n_genomes, uint64_t n_rec, uint32_t *nonuniform,
                              hipStream_t stream)
{
    if (n_genomes == 0) return hipSuccess;
    hipLaunchKernelGGL(rec_uniform_kernel, per_record_grid(n_genomes, n_rec), dim3(256), 0, stream, genomes, rec_off, n_genomes, nonuniform);
    return hipGetLastError();
}

hipError_t launch_brk_bytes(const GenomeDesc *genomes, const uint64_t *rec_off, uint32_t n_genomes, uint64_t n_rec, const uint32_t *nonuniform,
                            uint32_t *brk_bytes, hipStream_t stream)
{
    if (n_genomes == 0) return hipSuccess;
    // every genome is taken by exactly one of the two forms (long_records()); a batch may hold both kinds
    const uint32_t gy = std::min(n_genomes, 1024u), gx = std::max(4u, std::min(256u, 8192u / gy));
    hipLaunchKernelGGL(brk_zero_kernel, dim3(gx, gy), dim3(256), 0, stream, genomes, n_genomes, nonuniform, brk_bytes);
    hipLaunchKernelGGL(brk_set_kernel, per_record_grid(n_genomes, n_rec), dim3(256), 0, stream, genomes, rec_off, n_genomes, nonuniform, brk_bytes);
    hipLaunchKernelGGL(brk_bytes_kernel, per_record_grid(n_genomes, n_rec), dim3(256), 0, stream, genomes, rec_off, n_genomes, nonuniform, brk_bytes);
    return hipGetLastError();
}

// every genome of the batch was ITEM_SOLE: nothing to reduce, only the k-mer census to add up
__global__ void __launch_bounds__(256) census_kernel(FinalizeArgs a, uint32_t n_genomes)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long tot = 0;
    if (g < n_genomes) {
        const uint32_t i0 = a.genome_item_begin[g], i1 = a.genome_item_begin[g + 1];
        const uint64_t L = a.nvalid[g], nk = L >= (uint64_t)a.k ? L - (uint64_t)a.k + 1 : 0;
        for (uint32_t it = i0; it < i1; ++it)
            if ((uint64_t)a.items[it].word_begin * 16 < nk) tot += a.item_kmers[it];
    }
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(a.kmer_counter, tot);
}

hipError_t launch_census(const FinalizeArgs &args, uint32_t n_genomes, hipStream_t stream)
{
    if (n_genomes == 0) return hipSuccess;
    hipLaunchKernelGGL(census_kernel, dim3((n_genomes + 255) / 256), dim3(256), 0, stream, args, n_genomes);
    return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeArgs &args, uint32_t n_genomes, hipStream_t stream)
{
    if (n_genomes == 0) return hipSuccess;
    // 4 image bytes per thread and pass; many genomes in a batch -> latency-bound unless the workgroups are wide
    const uint32_t nwords = (args.algo == 0 ? HMH_M * 2 : (1u << args.p)) >> 2;
    const uint32_t threads = nwords >= 4096 ? 1024u : nwords >= 1024 ? 512u : 256u;
    switch (args.algo) {
    case 0: hipLaunchKernelGGL(finalize_kernel<0>, dim3(n_genomes), dim3(threads), 0, stream, args); break;
    case 1: hipLaunchKernelGGL(finalize_kernel<1>, dim3(n_genomes), dim3(threads), 0, stream, args); break;
    case 2: hipLaunchKernelGGL(finalize_kernel<2>, dim3(n_genomes), dim3(threads), 0, stream, args); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lash
