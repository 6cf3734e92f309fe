// pack_kernels.hip — ASCII records -> filtered 2-bit stream + record-break bitmap (gfx950).
//
// Replaces, for a whole batch at once, the two per-record copies the reference makes before its k-mer loop:
//   filter_out_n(seqrec.seq())   /root/reference/src/utils.rs:459 -> 33-41  (delete every byte that is not an
//                                upper-case A C G T and JOIN the flanks)
//   KSeq::new(&seq, 2)           utils.rs:464 (kmerutils 2-bit packing, A=0 C=1 G=2 T=3)
// and remembers where each record's surviving bases begin, because k-mers never span records
// (a fresh KSeq per record, utils.rs:457-464) but DO span deleted characters.
//
// v1 structure: one 256-thread workgroup walks one genome front to back in 4 KiB tiles, carrying the running
// count of surviving bases and the not-yet-complete output word from tile to tile, so no inter-workgroup
// communication is needed.  Parallelism = number of genomes (>= ~1000 at the BASELINE configs).
#include <hip/hip_runtime.h>

#include "lash_kernels.h"

namespace lash {

constexpr int PACK_THREADS = 256;
constexpr int PACK_TILE = PACK_THREADS * 16;             // bytes per tile: one 16-byte load per lane

// Four ASCII bytes -> four 2-bit codes (byte 0 first, in bits 7:6 of the result) and a 4-bit validity mask.
// code = (c >> 1) & 3 maps A C T G -> 0 1 2 3; x ^ (x >> 1) swaps 2 and 3 to get kmerutils' A C G T = 0 1 2 3.
// A byte is valid iff it equals the letter its own code would decode to (exact zero-byte test, no LUT).
__device__ __forceinline__ void classify4(uint32_t w, uint32_t &codes8, uint32_t &valid4)
{
    const uint32_t x = (w >> 1) & 0x03030303u;
    uint32_t e = (x << 1) | 0x41414141u;                   // 'A' 'C' 'E' 'G'
    const uint32_t m = (x >> 1) & ~x & 0x01010101u;        // 1 where x == 2
    e ^= m | (m << 4);                                     // 'E' ^ 0x11 = 'T'
    const uint32_t z = w ^ e;                              // zero byte <=> valid
    const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
    const uint32_t v = (~nz & 0x80808080u) >> 7;           // 0x01 per valid byte
    const uint32_t c = x ^ ((x >> 1) & 0x01010101u);
    codes8 = (c * 0x40100401u) >> 24;                      // byte j -> bits 7-2j..6-2j
    valid4 = ((v * 0x01020408u) >> 24) & 0xFu;             // byte j -> bit j
}

struct Lane16 {
    uint32_t bits;    // surviving bases, first in bits 31:30
    uint32_t cnt;     // how many survive
    uint32_t vmask;   // bit j: byte j survives
};

__device__ __forceinline__ Lane16 classify16(const uint4 q, uint32_t keep)
{
    uint32_t c0, c1, c2, c3, v0, v1, v2, v3;
    classify4(q.x, c0, v0);
    classify4(q.y, c1, v1);
    classify4(q.z, c2, v2);
    classify4(q.w, c3, v3);
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

__device__ __forceinline__ uint4 load16_guarded(const uint8_t *seq_lo, const uint8_t *seq_hi, const uint8_t *p)
{
    // p is 16-byte aligned; [seq_lo, seq_hi) is the caller's buffer.  Interior chunks use one dwordx4 load.
    if (p >= seq_lo && p + 16 <= seq_hi) return *reinterpret_cast<const uint4 *>(p);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int j = 0; j < 16; ++j) {
        const uint8_t *b = p + j;
        if (b >= seq_lo && b < seq_hi) w[j >> 2] |= (uint32_t)(*b) << (8 * (j & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ void __launch_bounds__(PACK_THREADS) pack_genome_kernel(PackArgs a)
{
    __shared__ uint32_t stage[PACK_TILE / 16 + 4];          // tile output, assembled by LDS ORs
    __shared__ uint32_t recbits[PACK_TILE / 32];            // which bytes of the tile start a record
    __shared__ uint32_t wave_tot[PACK_THREADS / 64];

    const GenomeDesc gd = a.genomes[blockIdx.x];
    const uint8_t *gbeg = a.seq + gd.byte_off, *gend = gbeg + gd.byte_len;
    const uint8_t *abeg = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(gbeg) & ~(uintptr_t)15);
    uint32_t *__restrict__ dstw = a.words + gd.word_off;
    uint32_t *__restrict__ dbrk = a.brk + gd.brk_off;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    uint64_t prefix = 0;          // surviving bases before this tile (uniform)
    uint32_t carry_bits = 0;      // the (prefix & 15) bases not yet written, first in bits 31:30 (uniform)
    uint64_t cursor = gd.rec_begin;

    for (const uint8_t *tb = abeg; tb < gend; tb += PACK_TILE) {
        // ---- 0. reset staging ----
        for (uint32_t i = tid; i < PACK_TILE / 16 + 4; i += PACK_THREADS) stage[i] = 0;
        if (tid < PACK_TILE / 32) recbits[tid] = 0;
        __syncthreads();

        // ---- 1. mark record starts that fall into this tile ----
        const uint64_t tile_lo = (uint64_t)(tb - a.seq), tile_hi = tile_lo + PACK_TILE;   // may wrap below 0 only for tb < seq: guarded by rec_off >= byte_off
        for (;;) {
            const uint64_t r = cursor + tid;
            bool in = false;
            if (r < gd.rec_end) {
                const uint64_t ro = a.rec_off[r];
                if ((int64_t)(ro - tile_lo) < (int64_t)PACK_TILE) {
                    in = true;
                    const uint32_t off = (uint32_t)(ro - tile_lo);
                    atomicOr(&recbits[off >> 5], 1u << (off & 31));
                }
            }
            const int n_in = __syncthreads_count(in);
            cursor += (uint64_t)n_in;
            if (n_in < PACK_THREADS) break;
        }
        (void)tile_hi;

        // ---- 2. classify this lane's 16 bytes ----
        const uint8_t *p = tb + 16 * tid;
        uint32_t keep = 0xFFFFu;
        if (p < gbeg) { const long d = gbeg - p; keep = d >= 16 ? 0u : (0xFFFFu << d) & 0xFFFFu; }
        if (p + 16 > gend) { const long d = gend - p; keep &= d <= 0 ? 0u : (d >= 16 ? 0xFFFFu : ((1u << d) - 1u)); }
        Lane16 l16{0, 0, 0};
        if (keep) l16 = classify16(load16_guarded(a.seq, a.seq_end, p), keep);

        // ---- 3. exclusive scan of the survivor counts over the workgroup ----
        uint32_t inc = l16.cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t n = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += n;
        }
        if (lane == 63) wave_tot[wid] = inc;
        __syncthreads();
        uint32_t wave_base = 0, tile_cnt = 0;
#pragma unroll
        for (int i = 0; i < PACK_THREADS / 64; ++i) {
            const uint32_t t = wave_tot[i];
            if (i < (int)wid) wave_base += t;
            tile_cnt += t;
        }
        const uint32_t excl = wave_base + inc - l16.cnt;    // survivors of this tile before this lane

        // ---- 4. record-break bits (global, sparse): position of the first survivor at/after each record start ----
        const uint32_t rb = (recbits[tid >> 1] >> ((tid & 1) * 16)) & 0xFFFFu;
        uint32_t m = rb;
        while (m) {
            const uint32_t j = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            const uint64_t pos = prefix + excl + (uint32_t)__builtin_popcount(l16.vmask & ((1u << j) - 1u));
            atomicOr(dbrk + (pos >> 5), 1u << (pos & 31));
        }

        // ---- 5. assemble the tile's output words in LDS ----
        const uint32_t carry = (uint32_t)(prefix & 15);
        if (tid == 0 && carry) atomicOr(&stage[0], carry_bits);
        if (l16.cnt) {
            const uint32_t q = carry + excl, wi = q >> 4, sh = (q & 15) * 2;
            atomicOr(&stage[wi], l16.bits >> sh);
            if (sh) atomicOr(&stage[wi + 1], l16.bits << (32 - sh));
        }
        __syncthreads();

        // ---- 6. store the complete words, carry the rest ----
        const uint32_t total = carry + tile_cnt, nfull = total >> 4;       // nfull <= 256
        if (tid < nfull) dstw[(prefix >> 4) + tid] = stage[tid];
        carry_bits = stage[nfull];
        prefix += tile_cnt;
        __syncthreads();
    }
    if (tid == 0) {
        if (prefix & 15) dstw[prefix >> 4] = carry_bits;     // last, partial word (zero-padded)
        a.nvalid[blockIdx.x] = prefix;
    }
    // a few defined words after the end keep look-ahead reads of the sketch kernel deterministic
    if (tid < PAD_WORDS) dstw[((prefix + 15) >> 4) + tid] = 0;
}

hipError_t launch_pack(const PackArgs &args, uint32_t n_genomes, hipStream_t stream)
{
    if (n_genomes == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_genome_kernel, dim3(n_genomes), dim3(PACK_THREADS), 0, stream, args);
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
