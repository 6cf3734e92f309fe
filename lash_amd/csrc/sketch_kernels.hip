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
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lash_kernels.h"

namespace lash {

enum { KM_16 = 0, KM_LT16 = 1, KM_GT16 = 2 };

// ------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s)
{
    return __builtin_amdgcn_alignbit(hi, lo, s);        // ({hi,lo} >> (s & 31))[31:0]
}

// reverse complement of the 16 bases of one packed word (first base in bits 31:30 on both sides)
__device__ __forceinline__ uint32_t rcword(uint32_t x)
{
    uint32_t y = __builtin_bitreverse32(~x);            // groups reversed, bits inside each group swapped
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}

__device__ __forceinline__ uint32_t clz64_nz(uint32_t hi, uint32_t lo)
{
    // count leading zeros of {hi,lo}; v_ffbh_u32 returns 0xFFFFFFFF for 0, which min() discards
    uint32_t ch = hi ? (uint32_t)__builtin_clz(hi) : 0xFFFFFFFFu;
    uint32_t cl = lo ? (uint32_t)__builtin_clz(lo) + 32u : 64u;
    return ch < cl ? ch : cl;
}

// XXH3-128 of the 4 little-endian bytes of w (XXH3_len_4to8_128b, len = 4), seed folded into `bitflip`.
__device__ __forceinline__ void xxh3_128_4b(uint32_t w, uint64_t bitflip, uint64_t &lo, uint64_t &hi)
{
    const uint32_t a0 = w ^ (uint32_t)bitflip, a1 = w ^ (uint32_t)(bitflip >> 32);
    constexpr uint64_t C = XXH_PRIME64_1 + 16;           // PRIME64_1 + (len << 2)
    constexpr uint32_t c0 = (uint32_t)C, c1 = (uint32_t)(C >> 32);
    // 64 x 64 -> 128 as four v_mad_u64_u32
    uint64_t t = (uint64_t)a0 * c0;
    uint64_t u = (uint64_t)a1 * c0 + (t >> 32);
    uint64_t v = (uint64_t)a0 * c1 + (uint32_t)u;
    uint64_t h = (uint64_t)a1 * c1 + ((u >> 32) + (v >> 32));
    uint64_t l = (uint64_t)(uint32_t)t | (v << 32);
    h += l << 1;
    l ^= h >> 3;
    l ^= l >> 35;
    l *= XXH_PRIME_MX2;
    l ^= l >> 28;
    h ^= h >> 37;
    h *= XXH_PRIME_MX1;
    h ^= h >> 32;
    lo = l;
    hi = h;
}

// XXH3-64 of the 8 little-endian bytes of {v_hi,v_lo} (XXH3_len_4to8_64b, len = 8 -> XXH3_rrmxmx)
__device__ __forceinline__ uint64_t xxh3_64_8b(uint32_t v_lo, uint32_t v_hi, uint64_t bitflip)
{
    // input64 = input2 + (input1 << 32): the two halves trade places
    uint64_t h = (((uint64_t)v_lo << 32) | v_hi) ^ bitflip;
    h ^= ((h << 49) | (h >> 15)) ^ ((h << 24) | (h >> 40));
    h *= XXH_PRIME_MX2;
    h ^= (h >> 35) + 8;
    h *= XXH_PRIME_MX2;
    return h ^ (h >> 28);
}

// ------------------------------------------------------------------------------------------------------------
// register spaces: LDS (the normal case) or global memory (2^p too large for 160 KiB of LDS)
// ------------------------------------------------------------------------------------------------------------
struct LdsRegs {
    uint32_t *base;
    __device__ __forceinline__ void umax(uint32_t i, uint32_t v) const
    { (void)__hip_atomic_fetch_max(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ void bor(uint32_t i, uint32_t v) const
    { (void)__hip_atomic_fetch_or(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return base[i]; }
};
struct GlobalRegs {
    uint32_t *base;
    __device__ __forceinline__ void umax(uint32_t i, uint32_t v) const
    { if (v) (void)__hip_atomic_fetch_max(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void bor(uint32_t i, uint32_t v) const
    { if (v) (void)__hip_atomic_fetch_or(base + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ uint32_t get(uint32_t i) const
    { return __hip_atomic_load(base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

// ------------------------------------------------------------------------------------------------------------
// the three add_kmer rules.  `vm` is 0 or ~0: invalid k-mers degrade to max(x,0) / OR 0.
// ------------------------------------------------------------------------------------------------------------
template <int ALGO, bool XLOW, class Regs>
__device__ __forceinline__ void add_kmer(const Regs &regs, uint32_t c_lo, uint32_t c_hi, uint32_t vm,
                                         uint64_t bitflip, int p)
{
    if constexpr (ALGO == 0) {
        // utils.rs:395-398: Sketch::add_bytes_with_seed(&(masked as u32).to_le_bytes(), seed)
        (void)c_hi;                                      // k > 16: only the low 32 bits are hashed
        uint64_t lo, hi;
        xxh3_128_4b(c_lo, bitflip, lo, hi);
        const uint64_t x = XLOW ? lo : hi, y = XLOW ? hi : lo;
        const uint32_t xh = (uint32_t)(x >> 32), xl = (uint32_t)x;
        const uint32_t bucket = xh >> 18;                                    // x >> 50
        const uint32_t th = alignbit(xh, xl, 18);                            // ((x << 14) ^ 0x3FFF) high word
        const uint32_t tl = (xl << 14) | 0x3FFFu;                            // ... low word, never 0
        const uint32_t lz = clz64_nz(th, tl) + 1;                            // 1..=51
        const uint32_t reg = ((lz << 10) | ((uint32_t)y & 0x3FFu)) & vm;
        regs.umax(bucket, reg);
    } else if constexpr (ALGO == 1) {
        // utils.rs:411-413: push_hash64(xxh3_64(masked.to_le_bytes(), seed)): bucket = low p bits,
        // rho = 1 + leading zeros of the remaining 64-p bits = clz64(h | (2^p - 1)) + 1
        const uint64_t h = xxh3_64_8b(c_lo, c_hi, bitflip);
        const uint32_t pm = (1u << p) - 1u;
        const uint32_t j = (uint32_t)h & pm;
        const uint32_t rho = clz64_nz((uint32_t)(h >> 32), (uint32_t)h | pm) + 1;
        regs.umax(j, rho & vm);
    } else {
        // utils.rs:427-429: UltraLogLog::add(h): idx = top p bits, bit (nlz + p - 1) of the register's prefix
        // bitmap; sequential pack(unpack(old) | bit) == pack(OR of all bits), so OR now, pack later
        const uint64_t h = xxh3_64_8b(c_lo, c_hi, bitflip);
        const uint32_t hh = (uint32_t)(h >> 32), hl = (uint32_t)h;
        const uint32_t idx = hh >> (32 - p) >> 0;                            // p <= 26 < 32
        const uint64_t t = (h << p) | ((1ull << p) - 1ull);                  // ~(~h << p)
        const uint32_t nlz = clz64_nz((uint32_t)(t >> 32), (uint32_t)t);    // 0..=64-p
        const uint32_t bit = nlz + (uint32_t)p - 1u;                         // p-1..=63
        (void)hl;
        regs.bor(idx * 2u + (bit >> 5), (1u << (bit & 31u)) & vm);
    }
}

// ------------------------------------------------------------------------------------------------------------
// which k-mer start positions of a lane's 64 are real k-mers of the reference's iterator?
// position i is valid iff i + k <= L (genome end) and no record begins in (i, i+k-1].
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t kmer_valid_mask(const uint32_t *bk, uint64_t pos0, uint64_t nk, int k)
{
    const uint64_t lim = nk - pos0;                       // caller guarantees pos0 < nk
    const uint64_t kvm = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
    const uint64_t wi = pos0 >> 5;                        // pos0 is a multiple of 64
    const uint32_t b0 = bk[wi], b1 = bk[wi + 1], b2 = bk[wi + 2];
    if ((b0 | b1 | b2) == 0u || k == 1) return kvm;
    // S(i) = OR_{d=1..k-1} B(i+d) by doubling on the 96-bit window
    uint64_t lo = (uint64_t)b0 | ((uint64_t)b1 << 32), hi = b2;
    lo = (lo >> 1) | (hi << 63);
    hi >>= 1;
    const int n = k - 1;
    int len = 1;
    while (len * 2 <= n) {
        lo |= (lo >> len) | (hi << (64 - len));
        hi |= hi >> len;
        len *= 2;
    }
    if (len < n) {
        const int s = n - len;
        lo |= (lo >> s) | (hi << (64 - s));
    }
    return kvm & ~lo;
}

// ------------------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------------------
template <int ALGO, int KMODE, bool XLOW, bool USE_LDS>
__global__ void __launch_bounds__(1024) sketch_kernel(SketchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_regs[];
    __shared__ unsigned long long kmer_count_wg;          // NB: static LDS object after the dynamic array is fine here (no LDS-DMA)

    const WorkItem it = a.items[blockIdx.x];
    const GenomeDesc gd = a.genomes[it.genome];
    const uint64_t L = a.nvalid[it.genome];
    const int k = a.k, p = a.p;
    const uint64_t nk = L >= (uint64_t)k ? L - (uint64_t)k + 1 : 0;     // k-mer start positions of the genome
    if ((uint64_t)it.word_begin * 16 >= nk) return;                       // slice beyond the surviving bases

    using Regs = typename std::conditional<USE_LDS, LdsRegs, GlobalRegs>::type;
    Regs regs;
    if constexpr (USE_LDS) {
        regs.base = lds_regs;
        for (uint32_t i = threadIdx.x; i < a.nreg32; i += blockDim.x) lds_regs[i] = 0;
    } else {
        regs.base = a.gregs + (uint64_t)blockIdx.x * a.nreg32;           // zeroed by the host (hipMemsetAsync)
    }
    if (threadIdx.x == 0) kmer_count_wg = 0;
    __syncthreads();

    const uint32_t *__restrict__ w = a.words + gd.word_off;
    const uint32_t *__restrict__ bk = a.brk + gd.brk_off;
    const uint64_t bitflip = a.bitflip;
    const uint32_t sh_lt = 32u - 2u * (uint32_t)k;                        // KM_LT16: fwd >>= sh_lt
    const uint32_t mask_lt = (KMODE == KM_LT16) ? ((1u << (2 * k)) - 1u) : 0xFFFFFFFFu;
    const uint32_t sh_gt = 64u - 2u * (uint32_t)k;                        // KM_GT16: fwd64 >>= sh_gt
    const uint64_t mask_gt = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint32_t my_kmers = 0;

    const uint32_t step = blockDim.x * SKETCH_WORDS_PER_THREAD;
    for (uint32_t tile = it.word_begin; tile < it.word_end; tile += step) {
        const uint32_t w0 = tile + threadIdx.x * SKETCH_WORDS_PER_THREAD;
        const uint64_t pos0 = (uint64_t)w0 * 16;
        if (w0 >= it.word_end || pos0 >= nk) continue;

        const uint4 q = *reinterpret_cast<const uint4 *>(w + w0);         // 64 bases, 16 B, coalesced
        uint32_t c0 = q.x, c1 = q.y, c2 = q.z, c3 = q.w;
        uint32_t c4 = w[w0 + 4];                                          // look-ahead (same or next cache line)
        uint32_t c5 = (KMODE == KM_GT16) ? w[w0 + 5] : 0u;
        uint64_t kv = kmer_valid_mask(bk, pos0, nk, k);
        my_kmers += (uint32_t)__builtin_popcountll(kv);

        uint32_t r0 = rcword(c0), r1 = rcword(c1), r2 = (KMODE == KM_GT16) ? rcword(c2) : 0u;
#pragma unroll 1
        for (int wi = 0; wi < SKETCH_WORDS_PER_THREAD; ++wi) {
            const uint32_t kvw = (uint32_t)kv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t vm = (uint32_t)__builtin_amdgcn_sbfe((int)kvw, r, 1);    // 0 or ~0
                uint32_t can_lo, can_hi = 0;
                if constexpr (KMODE == KM_GT16) {
                    const uint32_t fh = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
                    const uint32_t fl = r ? alignbit(c1, c2, 32 - 2 * r) : c1;
                    const uint64_t fwd = (((uint64_t)fh << 32) | fl) >> sh_gt;
                    const uint32_t rl = r ? alignbit(r1, r0, 2 * r) : r0;
                    const uint32_t rh = r ? alignbit(r2, r1, 2 * r) : r1;
                    const uint64_t rc = (((uint64_t)rh << 32) | rl) & mask_gt;
                    const uint64_t can = fwd < rc ? fwd : rc;                            // km.min(rc), utils.rs:494
                    can_lo = (uint32_t)can;
                    can_hi = (uint32_t)(can >> 32);
                } else {
                    uint32_t fwd = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
                    uint32_t rc = r ? alignbit(r1, r0, 2 * r) : r0;
                    if constexpr (KMODE == KM_LT16) { fwd >>= sh_lt; rc &= mask_lt; }
                    can_lo = fwd < rc ? fwd : rc;                                        // utils.rs:470,482
                }
                add_kmer<ALGO, XLOW>(regs, can_lo, can_hi, vm, bitflip, p);
            }
            // rotate the window by one word
            c0 = c1; c1 = c2; c2 = c3; c3 = c4; c4 = c5; c5 = 0;
            r0 = r1;
            if constexpr (KMODE == KM_GT16) { r1 = r2; r2 = rcword(c2); } else { r1 = rcword(c1); }
            kv >>= 16;
        }
    }

    // valid k-mer census (tests compare it with the oracle's iterator count)
    for (int off = 32; off > 0; off >>= 1) my_kmers += __shfl_down(my_kmers, off, 64);
    if ((threadIdx.x & 63) == 0 && my_kmers) atomicAdd(&kmer_count_wg, (unsigned long long)my_kmers);
    if constexpr (!USE_LDS) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && kmer_count_wg) atomicAdd(a.kmer_counter, kmer_count_wg);

    // flush the partial sketch in image register format (u16 LE for HMH, u8 for HLL / ULL)
    uint32_t *out = reinterpret_cast<uint32_t *>(a.partials + (uint64_t)blockIdx.x * a.partial_stride);
    if constexpr (ALGO == 0) {
        for (uint32_t i = threadIdx.x; i < HMH_M / 2; i += blockDim.x)
            out[i] = regs.get(2 * i) | (regs.get(2 * i + 1) << 16);
    } else if constexpr (ALGO == 1) {
        const uint32_t nw = (1u << p) >> 2;
        for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x)
            out[i] = regs.get(4 * i) | (regs.get(4 * i + 1) << 8) | (regs.get(4 * i + 2) << 16) | (regs.get(4 * i + 3) << 24);
    } else {
        const uint32_t nw = (1u << p) >> 2;
        for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t lo = regs.get(8 * i + 2 * b), hi = regs.get(8 * i + 2 * b + 1);
                uint32_t r = 0;
                if (lo | hi) {
                    // hash4j pack(): r = 4 * (index of leading one) + the two bits below it
                    const uint64_t x = ((uint64_t)hi << 32) | lo;
                    const uint32_t top = 63u - (uint32_t)__builtin_clzll(x);
                    const uint32_t below = top >= 2 ? (uint32_t)(x >> (top - 2)) & 3u : (uint32_t)(x << (2 - top)) & 3u;
                    r = (top << 2) | below;
                }
                o |= r << (8 * b);
            }
            out[i] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// finalize: reduce a genome's partial sketches (max / ULL merge), add the image header, optionally union into
// what is already in the image (LASH_F_ACCUMULATE, lash_merge_images).  One 256-thread workgroup per genome.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ull_unpack32pair(uint32_t r, uint32_t &hi)
{
    // hash4j unpack(): (4 | (r & 3)) << ((r >> 2) - 2); r == 0 -> 0.  Returns low word, hi by reference.
    if (r < 8) { hi = 0; return 0; }
    const uint64_t x = (uint64_t)(4u | (r & 3u)) << ((r >> 2) - 2u);
    hi = (uint32_t)(x >> 32);
    return (uint32_t)x;
}
__device__ __forceinline__ uint32_t ull_merge_reg(uint32_t a, uint32_t b)
{
    if (a == 0) return b;
    if (b == 0) return a;
    uint32_t ah, bh;
    const uint32_t al = ull_unpack32pair(a, ah), bl = ull_unpack32pair(b, bh);
    const uint64_t x = (((uint64_t)(ah | bh)) << 32) | (al | bl);
    const uint32_t top = 63u - (uint32_t)__builtin_clzll(x);
    return (top << 2) | ((uint32_t)(x >> (top - 2)) & 3u);               // top >= 2 because r >= 8
}

__device__ __forceinline__ uint32_t load_u32_any(const uint8_t *p)
{
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) return *reinterpret_cast<const uint32_t *>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ void store_u32_any(uint8_t *p, uint32_t v)
{
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) { *reinterpret_cast<uint32_t *>(p) = v; return; }
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

template <int ALGO>
__device__ __forceinline__ uint32_t merge_word(uint32_t a, uint32_t b)
{
    if constexpr (ALGO == 0) {
        const uint32_t al = a & 0xFFFFu, ah = a >> 16, bl = b & 0xFFFFu, bh = b >> 16;
        return (al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16);
    } else {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = (a >> (8 * i)) & 0xFFu, y = (b >> (8 * i)) & 0xFFu;
            const uint32_t m = ALGO == 1 ? (x > y ? x : y) : ull_merge_reg(x, y);
            o |= m << (8 * i);
        }
        return o;
    }
}

template <int ALGO>
__global__ void __launch_bounds__(256) finalize_kernel(FinalizeArgs a)
{
    __shared__ uint32_t hist[72];
    const uint32_t g = blockIdx.x;
    const uint32_t i0 = a.genome_item_begin[g], i1 = a.genome_item_begin[g + 1];
    uint64_t nk = ~0ull;
    if (a.nvalid) { const uint64_t L = a.nvalid[g]; nk = L >= (uint64_t)a.k ? L - (uint64_t)a.k + 1 : 0; }
    const uint32_t hdr = ALGO == 0 ? 0u : ALGO == 1 ? 33u : 8u;
    const uint32_t nbytes = ALGO == 0 ? HMH_M * 2 : (1u << a.p);
    const uint32_t nwords = nbytes >> 2;                                   // p >= 3 -> at least 2 words
    uint8_t *img = a.images + (uint64_t)g * a.image_bytes;
    if (threadIdx.x < 72) hist[threadIdx.x] = 0;
    __syncthreads();

    for (uint32_t wi = threadIdx.x; wi < nwords; wi += blockDim.x) {
        uint32_t acc = a.accumulate ? load_u32_any(img + hdr + 4ull * wi) : 0u;
        for (uint32_t it = i0; it < i1; ++it) {
            if ((uint64_t)a.items[it].word_begin * 16 >= nk) continue;     // slice never ran (see sketch_kernel)
            const uint8_t *src = a.partials + (uint64_t)it * a.partial_stride + a.partial_base_off;
            acc = merge_word<ALGO>(acc, load_u32_any(src + 4ull * wi));
        }
        store_u32_any(img + hdr + 4ull * wi, acc);
        if constexpr (ALGO == 1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t rho = (acc >> (8 * b)) & 0xFFu;
                atomicAdd(&hist[rho < 71u ? rho : 71u], 1u);
            }
        }
    }
    if constexpr (ALGO == 1) {
        // streaming_algorithms HyperLogLog header: alpha f64, zero u64, sum f64, p u8, len u64.
        // zero and sum are recomputed from the final registers; sum = sum_j 2^-m[j] is exact in f64 here.
        __syncthreads();
        if (threadIdx.x == 0) {
            double sum = 0.0;
            for (int r = 66; r >= 0; --r) {
                if (hist[r]) sum += (double)hist[r] * __longlong_as_double((long long)(1023 - r) << 52);
            }
            const uint64_t zero = hist[0];
            const uint64_t sum_bits = (uint64_t)__double_as_longlong(sum);
            const uint64_t len = 1ull << a.p;
            for (int b = 0; b < 8; ++b) {
                img[b] = (uint8_t)(a.alpha_bits >> (8 * b));
                img[8 + b] = (uint8_t)(zero >> (8 * b));
                img[16 + b] = (uint8_t)(sum_bits >> (8 * b));
                img[25 + b] = (uint8_t)(len >> (8 * b));
            }
            img[24] = (uint8_t)a.p;
        }
    } else if constexpr (ALGO == 2) {
        if (threadIdx.x == 0) {
            const uint64_t len = 1ull << a.p;                              // bincode Vec<u8> length prefix (switch U4)
            for (int b = 0; b < 8; ++b) img[b] = (uint8_t)(len >> (8 * b));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------------------
SketchPlan make_sketch_plan(int algo, int k, int p, bool x_low)
{
    SketchPlan s{};
    s.algo = algo; s.k = k; s.p = p; s.x_low = x_low;
    if (algo == 0) { s.nreg32 = HMH_M; s.partial_bytes = HMH_M * 2; }
    else if (algo == 1) { s.nreg32 = 1u << p; s.partial_bytes = 1u << p; }
    else { s.nreg32 = 2u << p; s.partial_bytes = 1u << p; }
    s.partial_stride = (s.partial_bytes + 15u) & ~15u;
    s.lds_bytes = s.nreg32 * 4u;
    s.use_lds = s.lds_bytes <= 128u * 1024u;
    s.threads = (s.use_lds && s.lds_bytes > 64u * 1024u) ? 1024u : 512u;  // <=64 KiB: two workgroups per CU
    if (!s.use_lds) s.lds_bytes = 0;
    return s;
}

template <int ALGO, int KMODE, bool XLOW, bool USE_LDS>
static hipError_t launch_one(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream)
{
    auto kern = sketch_kernel<ALGO, KMODE, XLOW, USE_LDS>;
    if (plan.lds_bytes > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_items), dim3(plan.threads), plan.lds_bytes, stream, args);
    return hipGetLastError();
}

template <int ALGO, bool XLOW>
static hipError_t launch_kmode(const SketchPlan &plan, const SketchArgs &args, uint32_t n, hipStream_t s)
{
    const int km = plan.k == 16 ? KM_16 : plan.k < 16 ? KM_LT16 : KM_GT16;
    if (plan.use_lds) {
        if (km == KM_16) return launch_one<ALGO, KM_16, XLOW, true>(plan, args, n, s);
        if (km == KM_LT16) return launch_one<ALGO, KM_LT16, XLOW, true>(plan, args, n, s);
        return launch_one<ALGO, KM_GT16, XLOW, true>(plan, args, n, s);
    }
    if (km == KM_16) return launch_one<ALGO, KM_16, XLOW, false>(plan, args, n, s);
    if (km == KM_LT16) return launch_one<ALGO, KM_LT16, XLOW, false>(plan, args, n, s);
    return launch_one<ALGO, KM_GT16, XLOW, false>(plan, args, n, s);
}

hipError_t launch_sketch(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream)
{
    if (n_items == 0) return hipSuccess;
    switch (plan.algo) {
    case 0: return plan.x_low ? launch_kmode<0, true>(plan, args, n_items, stream)
                              : launch_kmode<0, false>(plan, args, n_items, stream);
    case 1: return launch_kmode<1, false>(plan, args, n_items, stream);
    case 2: return launch_kmode<2, false>(plan, args, n_items, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_finalize(const FinalizeArgs &args, uint32_t n_genomes, hipStream_t stream)
{
    if (n_genomes == 0) return hipSuccess;
    switch (args.algo) {
    case 0: hipLaunchKernelGGL(finalize_kernel<0>, dim3(n_genomes), dim3(256), 0, stream, args); break;
    case 1: hipLaunchKernelGGL(finalize_kernel<1>, dim3(n_genomes), dim3(256), 0, stream, args); break;
    case 2: hipLaunchKernelGGL(finalize_kernel<2>, dim3(n_genomes), dim3(256), 0, stream, args); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace lash
