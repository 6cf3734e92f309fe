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

// reverse complement of the 16 bases of one packed word (first base in bits 31:30 on both sides)
__device__ __forceinline__ uint32_t rcword(uint32_t x)
{
    uint32_t y = __builtin_bitreverse32(~x);            // groups reversed, bits inside each group swapped
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}

// v_ffbh_u32: leading zeros of x, 0xFFFFFFFF for x == 0
__device__ __forceinline__ uint32_t ffbh_u32(uint32_t x)
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t pm_of(int p) { return (1u << p) - 1u; }

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

}  // namespace lash
