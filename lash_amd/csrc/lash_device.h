// lash_device.h — gfx950 device helpers shared by the sketch kernel and tools/ubench.hip:
// funnel-shift k-mer windows and the XXH3 short-input closed forms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lash_common.h"

namespace lash {

// ------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s)
{
    return __builtin_amdgcn_alignbit(hi, lo, s);        // ({hi,lo} >> (s & 31))[31:0]
}

// reverse complement of the 16 bases of one packed word (first base in bits 31:30 on both sides).  Complementing a base
// is an XOR of its 2-bit code with code[A]^code[T] (== code[C]^code[G] for every code assignment): `cm` holds that value
// in all 16 groups (LayoutDev::comp_mask; ~x for kmerutils' A,C,G,T = 0,1,2,3).
__device__ __forceinline__ uint32_t rcword(uint32_t x, uint32_t cm = 0xFFFFFFFFu)
{
    uint32_t y = __builtin_bitreverse32(x ^ cm);        // groups reversed, bits inside each group swapped
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}

// v_ffbh_u32: leading zeros of x, 0xFFFFFFFF for x == 0
__device__ __forceinline__ uint32_t ffbh_u32(uint32_t x)
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// min of two 64-bit values as {hi, lo} words.  hipcc's form is v_cmp_lt_u64 -> vcc + two VOP2 v_cndmask_b32 ... vcc, and on gfx950
// a VOP2 v_cndmask_b32 reading vcc holds the SIMD for 17-23 cycles (tools/ubench_isa: 12.7 cycles per instruction for the triple,
// 4.5 with the lane mask in a scalar pair and the VOP3 encoding of the select): a third of the k > 16 window's cost.
__device__ __forceinline__ void min_u64(uint32_t a_lo, uint32_t a_hi, uint32_t b_lo, uint32_t b_hi, uint32_t &lo, uint32_t &hi)
{
    uint64_t m;                                                           // lanes where a < b (a scalar pair, not vcc)
    const uint64_t a = ((uint64_t)a_hi << 32) | a_lo, b = ((uint64_t)b_hi << 32) | b_lo;
    asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(b_lo), "v"(a_lo), "s"(m));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(b_hi), "v"(a_hi), "s"(m));
}
__device__ __forceinline__ uint32_t pm_of(int p) { return (1u << p) - 1u; }

__device__ __forceinline__ uint32_t clz64_nz(uint32_t hi, uint32_t lo)
{
    // count leading zeros of {hi,lo}; v_ffbh_u32 returns 0xFFFFFFFF for 0, which min() discards
    uint32_t ch = hi ? (uint32_t)__builtin_clz(hi) : 0xFFFFFFFFu;
    uint32_t cl = lo ? (uint32_t)__builtin_clz(lo) + 32u : 64u;
    return ch < cl ? ch : cl;
}

// The seed-folded xxh3 constant as the two 32-bit words the hash xors into its input.  Which register file they sit in decides
// what that xor costs: gfx950 issues `v_xor_b32 v,v` (and literal / inline forms) in 2.6 cycles per wave and SIMD, `v_xor_b32 s,v`
// in 4.4 (tools/ubench_isa, profiles/r04/isa_cost/): the hot kernels keep the words in vector registers (vector()), code that
// runs once per junction or per drained k-mer takes them as they come (scalar()).
struct BitFlip {
    uint32_t lo, hi;
    static __device__ __forceinline__ BitFlip scalar(uint64_t b) { return BitFlip{(uint32_t)b, (uint32_t)(b >> 32)}; }
    static __device__ __forceinline__ BitFlip vector(uint64_t b)
    {
#ifdef LASH_SCALAR_CONSTS   // A/B build (tools/variants.sh): the words stay scalar
        return scalar(b);
#endif
        BitFlip f;
        asm volatile("v_mov_b32 %0, %1" : "=v"(f.lo) : "s"((uint32_t)b));     // (opaque: hipcc would fold a plain copy back into the scalar operand)
        asm volatile("v_mov_b32 %0, %1" : "=v"(f.hi) : "s"((uint32_t)(b >> 32)));
        return f;
    }
};

// 64 x 64 -> 128 multiply of {a1,a0} by the XXH3 constant PRIME64_1 + (len << 2), len = 4.
// Four v_mad_u64_u32; the third one adds the full 64-bit partial sum and its carry-out (VCC) is folded into the
// fourth's addend, which saves three register-pair moves and a 64-bit add over the textbook chain:
//   t = a0*c0;  u = a1*c0 + hi32(t);  {carry, v} = a0*c1 + u;  hi = a1*c1 + {carry, hi32(v)};  lo = {lo32(v), lo32(t)}
__device__ __forceinline__ void xxh3_mul128(uint32_t a0, uint32_t a1, uint64_t &lo, uint64_t &hi)
{
    constexpr uint64_t C = XXH_PRIME64_1 + 16;
    constexpr uint32_t c0 = (uint32_t)C, c1 = (uint32_t)(C >> 32);
    const uint64_t t = (uint64_t)a0 * c0;
    const uint64_t u = (uint64_t)a1 * c0 + (t >> 32);
    uint64_t v;
    uint32_t carry;
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
        : "=v"(v), "=v"(carry)
        : "v"(a0), "s"(c1), "v"(u)
        : "vcc");
    hi = (uint64_t)a1 * c1 + (((uint64_t)carry << 32) | (v >> 32));
    lo = (uint64_t)(uint32_t)t | (v << 32);
}

// XXH3-128 of the 4 little-endian bytes of w (XXH3_len_4to8_128b, len = 4), seed folded into `bitflip`.
__device__ __forceinline__ void xxh3_128_4b(uint32_t w, BitFlip bitflip, uint64_t &lo, uint64_t &hi)
{
    uint64_t l, h;
    xxh3_mul128(w ^ bitflip.lo, w ^ bitflip.hi, l, h);
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

// The part of the same hash that a HyperMinHash update with x = high64 needs when the rank fits in 18 bits:
//   xh    = bits 63:32 of the high half (bucket = xh >> 18, rank field = xh & 0x3FFFF)
//   sig10 = bits 9:0 of the low half
// Skips the low word of the last multiply and the final xorshift of the high half (they only feed bits 31:0 of x).
__device__ __forceinline__ void xxh3_128_4b_hmh_fast(uint32_t w, BitFlip bitflip, uint32_t &xh, uint32_t &sig10)
{
    uint64_t l, h;
    xxh3_mul128(w ^ bitflip.lo, w ^ bitflip.hi, l, h);
    h += l << 1;
    l ^= h >> 3;
    l ^= l >> 35;
    // (the two xorshifts in hand-written 32-bit halves — alignbit + xor3 — were tried: hipcc then folds the doubling above back
    // into the multiply chain, or pays an s_nop for the pair hazard: 41 instead of 37 instructions per k-mer.  Left as is.)
    // sig10 = bits 9:0 of (m ^ m >> 28), m = l * MX2: only bits 37:0 of the product matter.  Bits 31:0 come from
    // l_lo * MX2_lo; bits 37:32 are the low 6 bits of hi32(l_lo * MX2_lo) + l_lo * MX2_hi + l_hi * MX2_lo, and both halves of
    // MX2 end in the same six bits (0x25), so the two cross products collapse into ONE 24-bit multiply:
    // 37 * (l_lo + l_hi).  Two v_mul_lo_u32 + v_add3 become v_add + v_mad_u32_u24 (-6 of ~146 cycles per k-mer).
    static_assert(((uint32_t)XXH_PRIME_MX2 & 63u) == 37u && ((uint32_t)(XXH_PRIME_MX2 >> 32) & 63u) == 37u, "low six bits of both halves");
    const uint64_t pm = (uint64_t)(uint32_t)l * (uint32_t)XXH_PRIME_MX2;
    uint32_t top;                                                                 // bits 5:0 valid
    asm("v_mad_u32_u24 %0, %1, 37, %2" : "=v"(top) : "v"((uint32_t)l + (uint32_t)(l >> 32)), "v"((uint32_t)(pm >> 32)));
    sig10 = ((uint32_t)pm ^ alignbit(top, (uint32_t)pm, 28)) & 0x3FFu;
    h ^= h >> 37;
    constexpr uint32_t m0 = (uint32_t)XXH_PRIME_MX1, m1 = (uint32_t)(XXH_PRIME_MX1 >> 32);
    const uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
    xh = __umulhi(h0, m0) + h0 * m1 + h1 * m0;
}

// ... and only the half that decides bucket and rank (the signature half — a third of the instructions — is left to the few
// k-mers whose rank can still win their bucket: process_word_defer)
__device__ __forceinline__ uint32_t xxh3_128_4b_hmh_rank(uint32_t w, BitFlip bitflip)
{
    uint64_t l, h;
    xxh3_mul128(w ^ bitflip.lo, w ^ bitflip.hi, l, h);
    asm("" : "+v"(l));                                   // (opaque: with nothing else using l, hipcc folds the doubling into the multiply chain — 8 instructions for one)
    asm("v_lshl_add_u64 %0, %1, 1, %0" : "+v"(h) : "v"(l));   // h += l << 1
    h ^= h >> 37;
    constexpr uint32_t m0 = (uint32_t)XXH_PRIME_MX1, m1 = (uint32_t)(XXH_PRIME_MX1 >> 32);
    const uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
    return __umulhi(h0, m0) + h0 * m1 + h1 * m0;
}

// ---- the same three forms for x = LOW half (layout.hmh_x_low, SURVEY App. D switch U1; round 6) ---------------------------------
// With the halves swapped the rank half is the `l` chain — l ^= h >> 3; l ^= l >> 35; l *= MX2; l ^= l >> 28 — and the signature the low
// ten bits of avalanche(h).  The l chain in 32-bit halves: only l_hi ^= h_hi >> 3 and l_lo ^= (h >> 3)_lo ^ (l_hi >> 3) are ever
// needed (five instructions), and the high word of the last multiply is the same three-multiply sum the default takes from h.
__device__ __forceinline__ uint32_t xxh3_l_chain_high(uint64_t l, uint64_t h)    // bits 63:32 of (l ^ h >> 3, xorshift 35) * MX2
{
    uint64_t hs;                                                            // h >> 3: ONE 64-bit shift (4.4 cycles) for a shift and a funnel shift (7.3)
    asm("v_lshrrev_b64 %0, 3, %1" : "=v"(hs) : "v"(h));
    const uint32_t l1 = (uint32_t)(l >> 32) ^ (uint32_t)(hs >> 32);
    const uint32_t l0 = __builtin_amdgcn_bitop3_b32((uint32_t)l, (uint32_t)hs, l1 >> 3, 0x96);
    constexpr uint32_t m0 = (uint32_t)XXH_PRIME_MX2, m1 = (uint32_t)(XXH_PRIME_MX2 >> 32);
    return __umulhi(l0, m0) + l0 * m1 + l1 * m0;
}
// xh = bits 63:32 of the low half (bucket = xh >> 18, rank field = xh & 0x3FFFF), sig10 = bits 9:0 of the high half
__device__ __forceinline__ void xxh3_128_4b_hmh_fast_xlow(uint32_t w, BitFlip bitflip, uint32_t &xh, uint32_t &sig10)
{
    uint64_t l, h;
    xxh3_mul128(w ^ bitflip.lo, w ^ bitflip.hi, l, h);
    h += l << 1;
    const uint32_t t = xxh3_l_chain_high(l, h);
    xh = t ^ (t >> 28);                                                     // the final xorshift reaches bits 3:0 of the word
    // sig10 = bits 9:0 of (m ^ m >> 32), m = (h ^ h >> 37) * MX1: bits 9:0 of both words of the product.  The low word and the carry
    // into the high one come from ONE v_mad_u64_u32; of the two cross products only ten bits matter: full-rate 24-bit multiplies
    const uint32_t g1 = (uint32_t)(h >> 32), g0 = (uint32_t)h ^ (g1 >> 5);
    const uint64_t pm = (uint64_t)g0 * (uint32_t)XXH_PRIME_MX1;
    uint32_t top;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(top) : "v"(g0), "s"((uint32_t)(XXH_PRIME_MX1 >> 32) & 0xFFFFFFu), "v"((uint32_t)(pm >> 32)));
    asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(top) : "v"(g1), "s"((uint32_t)XXH_PRIME_MX1 & 0xFFFFFFu));
    sig10 = ((uint32_t)pm ^ top) & 0x3FFu;
}
// ... and the rank half alone, WITHOUT the final xorshift: that one only moves bits 31:28 onto bits 3:0, and the deferring filter looks
// at bits 31:4 — the bucket (31:18) and the upper 14 of the 16 rank bits below it; LdsThrRegs caps its thresholds at 14 leading
// zeros in this variant so that the two bits it cannot trust never decide (see there)
__device__ __forceinline__ uint32_t xxh3_128_4b_hmh_rank_xlow(uint32_t w, BitFlip bitflip)
{
    uint64_t l, h;
    xxh3_mul128(w ^ bitflip.lo, w ^ bitflip.hi, l, h);
    asm("" : "+v"(l));                                   // (opaque, as in xxh3_128_4b_hmh_rank: or hipcc folds the doubling into the multiply chain —
    asm("v_lshl_add_u64 %0, %1, 1, %0" : "+v"(h) : "v"(l));   //  a fifth v_mad_u64_u32 and seven more instructions per k-mer)
    return xxh3_l_chain_high(l, h);
}

// XXH3-64 of the 8 little-endian bytes of {v_hi,v_lo} (XXH3_len_4to8_64b, len = 8 -> XXH3_rrmxmx), up to but NOT including the
// final `h ^= h >> 28`
__device__ __forceinline__ uint64_t xxh3_64_8b_pre(uint32_t v_lo, uint32_t v_hi, BitFlip bitflip)
{
    // input64 = input2 + (input1 << 32): the two halves trade places
    const uint32_t lo = v_hi ^ bitflip.lo, hi = v_lo ^ bitflip.hi;
    // h ^= rotl(h, 49) ^ rotl(h, 24) in 32-bit halves: rotl 49 = rotr 15, rotl 24 = halves swapped + rotr 8; every half of a rotated
    // value is one v_alignbit, the three-way xor one v_bitop3 — 6 instructions; as 64-bit shifts hipcc needs 10, two of them
    // v_lshrrev_b64 / v_lshlrev_b64 at 5.9 cycles
    const uint32_t nlo = __builtin_amdgcn_bitop3_b32(lo, alignbit(hi, lo, 15), alignbit(lo, hi, 8), 0x96);
    const uint32_t nhi = __builtin_amdgcn_bitop3_b32(hi, alignbit(lo, hi, 15), alignbit(hi, lo, 8), 0x96);
    uint64_t h = ((uint64_t)nhi << 32) | nlo;
    h *= XXH_PRIME_MX2;
    h ^= (h >> 35) + 8;
    h *= XXH_PRIME_MX2;
    return h;
}
__device__ __forceinline__ uint64_t xxh3_64_8b(uint32_t v_lo, uint32_t v_hi, BitFlip bitflip)
{
    const uint64_t h = xxh3_64_8b_pre(v_lo, v_hi, bitflip);
    return h ^ (h >> 28);
}

// ---- UltraLogLog registers (hash4j pack / unpack as ported by crate ultraloglog) ------------------------
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

// The same merge without 64-bit bitmaps (exhaustively equal for every pair of valid registers, 0 or 8..255; tools/ & tests):
// with hi >= lo, d = top(hi) - top(lo): lo's leading one and the bit below it land on hi's two low bits only when d <= 2 —
//   d = 0: hi | (lo & 3);   d = 1: hi | 2 | ((lo >> 1) & 1);   d = 2: hi | 1;   else hi      (table of 16 two-bit entries).
__device__ __forceinline__ uint32_t ull_merge_fast(uint32_t a, uint32_t b)
{
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    uint32_t d = (hi >> 2) - (lo >> 2);
    d = d < 3u ? d : 3u;
    d = lo ? d : 3u;                                          // an empty register adds nothing
    return hi | ((0x55FAE4u >> (2u * (4u * d + (lo & 3u)))) & 3u);
}

}  // namespace lash
