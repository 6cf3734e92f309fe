// dist_kernels.hip — pair statistics for `lash dist` on HyperMinHash and HyperLogLog sketches.
//
// Replaces the register scan inside hyperminhash's Sketch::similarity, called once per (reference, query) pair from
// /root/reference/src/utils.rs:164:   C = #{i : a_i != 0 and a_i == b_i},   N = #{i : a_i != 0 or b_i != 0}.
// The O(N_ref * N_qry * 16384) part runs here; cardinalities and the collision correction are O(pairs) host work.
//
// Both kernels tile the pair matrix and walk the registers in chunks staged in LDS (details at each kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "lash_device.h"
#include "lash_kernels.h"
#include "ull_estimators.h"

namespace lash {

constexpr int DT = 16;                 // HLL kernel: tile edge (pairs), one pair per lane

// Triangular calls (reference set == query set, utils.rs:158-160: only pairs with column <= row are printed).  `tri` is the
// set index of the call's first row (-1: rectangular call); a tile whose first column lies beyond its last row has nothing
// wanted in it and returns at once — its outputs stay unwritten.
__device__ __forceinline__ bool tile_above_diagonal(int64_t tri, uint32_t r0, uint32_t tile_rows, uint32_t q0)
{
    return tri >= 0 && (int64_t)q0 > tri + (int64_t)r0 + (int64_t)tile_rows - 1;
}

// ---- HyperMinHash: two u16 registers per u32, compared with packed 16-bit VALU ops --------------------------------------
// per half h of a word pair (a, b):   [a_h != 0 and a_h != b_h] = min(a_h ^ b_h, min(a_h, 1))
//                                     [a_h != 0 or  b_h != 0]   = min(a_h, 1) | min(b_h, 1)
// C = (non-zero registers of a) - sum of the first, N = sum of the second; the sums run in packed u16 counters (at most
// 8 192 words per sketch, so a half never exceeds 8 192).  A 256-thread workgroup owns a 64 x 64 tile of pairs, a lane a
// 4 x 4 block of it: the flags min(a,1) / min(b,1) are computed once per operand word and shared by 4 pairs, and one
// ds_read_b128 per operand side fetches the lane's 4 rows (LDS chunks are stored word-major: T[word][row]).
// Only the two min() need packed 16-bit ops (v_pk_min_u16, ~4.2 cycles); xor / or / add are plain 32-bit ops (~2.4).
// hipcc turns min(x, 1) on u16 vectors into compare + select per half, hence the inline asm.
constexpr int HT = 64;                 // tile edge (pairs)
constexpr int HB = 4;                  // block edge per lane
constexpr int HCW = 64;                // words (= 128 registers) per sketch per LDS chunk
constexpr int HSTRIDE = HT + 4;        // LDS row (one word of 64 sketches), padded, 16-byte multiple

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// Images: `hdr` bytes of header, then 16384 u16 registers; consecutive images `stride` bytes apart.  C and N only ask
// "zero?" and "equal?", so the byte order of the registers (layout.hmh_reg_be) does not matter here.
__device__ __forceinline__ uint32_t hmh_word(const uint8_t *img, uint32_t w)
{
    const uint8_t *p = img + 4ull * w;
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) return *reinterpret_cast<const uint32_t *>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

__global__ void __launch_bounds__(256) hmh_pairs_kernel(const uint8_t *__restrict__ ref, uint32_t n_ref,
                                                        const uint8_t *__restrict__ qry, uint32_t n_qry, uint32_t hdr,
                                                        uint64_t stride, uint32_t *__restrict__ out_c,
                                                        uint32_t *__restrict__ out_n, int64_t tri)
{
    __shared__ __attribute__((aligned(16))) uint32_t R[HCW][HSTRIDE], Q[HCW][HSTRIDE];
    const uint32_t tid = threadIdx.x, tr = tid / 16, tq = tid % 16;           // lane block: rows tr*4.., columns tq*4..
    const uint32_t r0 = blockIdx.y * HT, q0 = blockIdx.x * HT;
    if (tile_above_diagonal(tri, r0, HT, q0)) return;
    constexpr uint32_t WORDS = HMH_M / 2;                                       // 8192 u32 per sketch
    uint32_t one = 0x00010001u;
    asm volatile("" : "+v"(one));                                              // keep it in a VGPR (VOP3P operand)
    uint32_t acc_ne[HB][HB], acc_or[HB][HB], nz_a[HB];                          // packed u16 counters
#pragma unroll
    for (int i = 0; i < HB; ++i) {
        nz_a[i] = 0;
#pragma unroll
        for (int j = 0; j < HB; ++j) { acc_ne[i][j] = 0; acc_or[i][j] = 0; }
    }
    for (uint32_t w0 = 0; w0 < WORDS; w0 += HCW) {
        // stage: global reads run along the words of a sketch (coalesced), LDS is written word-major
        for (uint32_t i = tid; i < HT * HCW; i += 256) {
            const uint32_t row = i / HCW, col = i % HCW;
            R[col][row] = (r0 + row < n_ref) ? hmh_word(ref + (uint64_t)(r0 + row) * stride + hdr, w0 + col) : 0u;
            Q[col][row] = (q0 + row < n_qry) ? hmh_word(qry + (uint64_t)(q0 + row) * stride + hdr, w0 + col) : 0u;
        }
        __syncthreads();
#pragma unroll 2
        for (uint32_t w = 0; w < HCW; ++w) {
            const uint4 av = *reinterpret_cast<const uint4 *>(&R[w][tr * HB]);
            const uint4 bv = *reinterpret_cast<const uint4 *>(&Q[w][tq * HB]);
            const uint32_t a[HB] = {av.x, av.y, av.z, av.w}, b[HB] = {bv.x, bv.y, bv.z, bv.w};
            uint32_t fa[HB], fb[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) { fa[i] = pk_min_u16(a[i], one); fb[i] = pk_min_u16(b[i], one); nz_a[i] += fa[i]; }
#pragma unroll
            for (int i = 0; i < HB; ++i) {
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    acc_ne[i][j] += pk_min_u16(a[i] ^ b[j], fa[i]);       // plain adds: a half never exceeds 8 192
                    acc_or[i][j] += fa[i] | fb[j];
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < HB; ++i) {
        const uint32_t r = r0 + tr * HB + i;
        const uint32_t nza = (nz_a[i] & 0xFFFFu) + (nz_a[i] >> 16);
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            const uint32_t q = q0 + tq * HB + j;
            if (r < n_ref && q < n_qry) {
                out_c[(uint64_t)r * n_qry + q] = nza - ((acc_ne[i][j] & 0xFFFFu) + (acc_ne[i][j] >> 16));
                out_n[(uint64_t)r * n_qry + q] = (acc_or[i][j] & 0xFFFFu) + (acc_or[i][j] >> 16);
            }
        }
    }
}

// ---- HyperMinHash: the expected-collision term of every pair of SMALL sketches -------------------------------------------
// hyperminhash's similarity subtracts expected_collisions(n, m) from the matching-register count (utils.rs:164 behind
// Sketch::similarity).  For cardinalities above 2^(p+5) that is a closed form; below it — viruses, plasmids, short contigs —
// the crate walks all 2^q x 2^r = 65 536 (leading-zero, signature) cells:
//     x = sum_cells [ (1-b2)^n - (1-b1)^n ] * [ (1-b2)^m - (1-b1)^m ],   b1 < b2 the cell's hash-value interval,
// 262 144 pow() per PAIR: 4 ms to 0.2 s of host time each (measured), i.e. days for a 10^3 x 10^3 collection — the same in
// the reference.  The cell factors depend on one cardinality only, so every sketch gets its vector of 65 536 cell
// probabilities once (collision_vectors_kernel) and the pair sums are the f64 matrix product of the reference vectors with
// the query vectors (collision_gemm_kernel, v_mfma_f64_16x16x4_f64).  The sum runs in another order than the crate's
// loop: results agree to ~1e-13 relative, far inside the 6 decimals `dist` prints.
constexpr int EC_CELLS = 65536;              // 64 x 1024
__global__ void __launch_bounds__(256) collision_vectors_kernel(const double *__restrict__ card, double *__restrict__ P)
{
    const uint32_t cell = blockIdx.x * 256u + threadIdx.x, s = blockIdx.y;
    const int i = (int)(cell >> 10) + 1;                        // 1..64
    const double j = (double)((cell & 1023u) + 1u);             // 1..1024
    double b1, b2;
    if (i != 64) {
        const double den = ldexp(1.0, HMH_P + 10 + i);
        b1 = (1024.0 + j) / den;
        b2 = (1024.0 + j + 1.0) / den;
    } else {
        const double den = ldexp(1.0, HMH_P + 10 + i - 1);
        b1 = j / den;
        b2 = (j + 1.0) / den;
    }
    const double c = card[s];
    P[(uint64_t)s * EC_CELLS + cell] = pow(1.0 - b2, c) - pow(1.0 - b1, c);
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

// X[M][N] = A[M][65536] * B[N][65536]^T.  One workgroup = a 128 x 128 tile of X; its four waves own 64 x 64 quarters as
// 4 x 4 MFMA blocks (16 flop per byte of vector data read; 64 x 64 tiles at 8 flop/B were bound by re-reading the vectors:
// 2.9 TB/s, 25 TFLOP/s).  The next 16 cells of the tile's 256 rows are fetched into registers while the current 16 are
// multiplied out of LDS.  Tiles are numbered so that the workgroups an XCD receives (every 8th) form one band of rows: they
// share A through that XCD's L2.
// Operand layout of v_mfma_f64_16x16x4_f64: A lane l = A[row l&15][k l>>4], B lane l = B[k l>>4][col l&15],
// D register i of lane l = D[row (l>>4) + 4i][col l&15].
constexpr int EC_T = 128;                    // tile edge
constexpr int EC_KC = 16;                    // cells per LDS stage
constexpr int EC_LD = EC_KC + 1;             // padded LDS row (doubles)

__global__ void __launch_bounds__(256) collision_gemm_kernel(const double *__restrict__ A, uint32_t M, const double *__restrict__ B,
                                                             uint32_t N, double *__restrict__ X, uint32_t tiles_x, uint32_t n_tiles)
{
    __shared__ double As[EC_T][EC_LD], Bs[EC_T][EC_LD];
    const uint32_t per_xcd = (n_tiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t r0 = (tile / tiles_x) * EC_T, q0 = (tile % tiles_x) * EC_T;
    const uint32_t wr = (wave >> 1) * 64u, wq = (wave & 1u) * 64u;
    v4f64 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = v4f64{0.0, 0.0, 0.0, 0.0};
    // staging: thread t moves 8 consecutive cells of row t/2 (of A and of B): 2 threads x 8 cells = the 16 cells of a stage
    const uint32_t lrow = tid >> 1, lk = (tid & 1u) * 8u;
    const bool a_ok = r0 + lrow < M, b_ok = q0 + lrow < N;
    const double *ap = A + (uint64_t)(a_ok ? r0 + lrow : 0u) * EC_CELLS + lk;
    const double *bp = B + (uint64_t)(b_ok ? q0 + lrow : 0u) * EC_CELLS + lk;
    double pa[8], pb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { pa[u] = a_ok ? ap[u] : 0.0; pb[u] = b_ok ? bp[u] : 0.0; }
    for (uint32_t k0 = 0; k0 < (uint32_t)EC_CELLS; k0 += EC_KC) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { As[lrow][lk + u] = pa[u]; Bs[lrow][lk + u] = pb[u]; }
        __syncthreads();
        if (k0 + EC_KC < (uint32_t)EC_CELLS) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { pa[u] = a_ok ? ap[k0 + EC_KC + u] : 0.0; pb[u] = b_ok ? bp[k0 + EC_KC + u] : 0.0; }
        }
#pragma unroll
        for (int kk = 0; kk < EC_KC; kk += 4) {
            const uint32_t kc = kk + (lane >> 4), rr = lane & 15u;
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = As[wr + 16u * a + rr][kc]; bv[a] = Bs[wq + 16u * a + rr][kc]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t r = r0 + wr + 16u * a + (lane >> 4) + 4u * i, q = q0 + wq + 16u * b + (lane & 15u);
                if (r < M && q < N) X[(uint64_t)r * N + q] = acc[a][b][i];
            }
}

// ---- HyperLogLog: union statistics of every pair (utils.rs:355-363: ref_hll.union(q_hll); ref_hll.len()) ----------------
// union = register-wise max; len() needs only zero = #{max == 0} and sum = sum_i 2^-max_i of the union, so no union
// sketch is ever materialised.  sum is accumulated exactly as two integers (registers <= 32: units of 2^-32; larger
// ones: units of 2^-64) and rounded once.  Images are `33-byte header + 2^p register bytes` (finalize_kernel), so the
// register arrays sit at odd addresses: staged into LDS byte-wise (the whole problem is L2-resident).
constexpr int HCHUNK = 1024;                 // registers per sketch per chunk
constexpr int HROW = HCHUNK / 4 + 1;         // LDS row in words, padded

__global__ void __launch_bounds__(256) hll_pairs_kernel(const uint8_t *__restrict__ ref, uint32_t n_ref,
                                                        const uint8_t *__restrict__ qry, uint32_t n_qry, int p, uint32_t hdr,
                                                        uint32_t *__restrict__ out_zero, double *__restrict__ out_sum, int64_t tri)
{
    __shared__ uint32_t R[DT][HROW], Q[DT][HROW];
    const uint32_t tid = threadIdx.x, tr = tid / DT, tq = tid % DT;
    const uint32_t r0 = blockIdx.y * DT, q0 = blockIdx.x * DT;
    if (tile_above_diagonal(tri, r0, DT, q0)) return;
    const uint32_t m = 1u << p;
    const uint64_t stride = (uint64_t)hdr + m;
    unsigned long long s1 = 0, s2 = 0;          // sum of 2^(32-r) over r <= 32 ; sum of 2^(64-r) over r > 32
    uint32_t zero = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += HCHUNK) {
        const uint32_t n = m - c0 < (uint32_t)HCHUNK ? m - c0 : (uint32_t)HCHUNK;     // m >= 16: a multiple of 4
        for (uint32_t i = tid; i < DT * (n / 4); i += 256) {
            const uint32_t row = i / (n / 4), col = i % (n / 4);
            uint32_t a = 0, b = 0;
            if (r0 + row < n_ref) {
                const uint8_t *s = ref + (uint64_t)(r0 + row) * stride + hdr + c0 + 4 * col;
                a = s[0] | (s[1] << 8) | (s[2] << 16) | ((uint32_t)s[3] << 24);
            }
            if (q0 + row < n_qry) {
                const uint8_t *s = qry + (uint64_t)(q0 + row) * stride + hdr + c0 + 4 * col;
                b = s[0] | (s[1] << 8) | (s[2] << 16) | ((uint32_t)s[3] << 24);
            }
            R[row][col] = a;
            Q[row][col] = b;
        }
        __syncthreads();
        for (uint32_t w = 0; w < n / 4; ++w) {
            const uint32_t a = R[tr][w], b = Q[tq][w];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t x = (a >> (8 * j)) & 0xFFu, y = (b >> (8 * j)) & 0xFFu;
                const uint32_t r = x > y ? x : y;
                zero += r == 0u;
                if (r <= 32u) s1 += 1ull << (32u - r);
                else s2 += r < 64u ? 1ull << (64u - r) : (r == 64u ? 1ull : 0ull);   // rho <= 64 - p + 1 <= 61 in practice
            }
        }
        __syncthreads();
    }
    if (r0 + tr < n_ref && q0 + tq < n_qry) {
        const uint64_t o = (uint64_t)(r0 + tr) * n_qry + q0 + tq;
        out_zero[o] = zero;
        out_sum[o] = (double)s1 * 2.3283064365386963e-10 + (double)s2 * 5.421010862427522e-20;   // 2^-32, 2^-64
    }
}


// ---- HyperLogLog pair statistics through threshold bitmaps (p >= 10) ------------------------------------------------------
// With A_t = {i : a_i <= t} as a bitmap, #{i : max(a_i, b_i) <= t} = popcount(A_t & B_t) =: c_t, and
//     sum_i 2^-max(a_i, b_i) = sum_{t < hi} c_t 2^-(t+1) + m 2^-hi,      zero = c_0,
// where [lo, hi] is the range of register values in the call (a sketch's registers sit in a band of ~20 values around
// log2(n / m)).  Per pair that is band x m/32 (AND + popcount-accumulate) instead of m x ~10 byte operations: 7x fewer
// instructions at p = 14.  The sum is formed from the same exact integers as in hll_pairs_kernel (units of 2^-32 and 2^-64,
// rounded once), so both kernels return identical bits.
__global__ void __launch_bounds__(256) hll_minmax_kernel(const uint8_t *__restrict__ img, uint32_t hdr, uint64_t stride, uint32_t m,
                                                         uint32_t *__restrict__ lohi)
{
    // one workgroup per sketch, four consecutive bytes per lane and step (the rows sit at odd addresses); two atomics per
    // workgroup — per-wave atomics on the two words serialised into 2 ms for 2 048 sketches
    __shared__ uint32_t wlo[4], whi[4];
    const uint8_t *row = img + (uint64_t)blockIdx.x * stride + hdr;
    uint32_t lo = 255u, hi = 0u;
    for (uint32_t at = threadIdx.x * 4u; at < m; at += 1024u) {
        const uint32_t a = row[at], b = row[at + 1], c = row[at + 2], d = row[at + 3];
        const uint32_t l1 = a < b ? a : b, l2 = c < d ? c : d, h1 = a > b ? a : b, h2 = c > d ? c : d;
        lo = l1 < lo ? l1 : lo; lo = l2 < lo ? l2 : lo;
        hi = h1 > hi ? h1 : hi; hi = h2 > hi ? h2 : hi;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63u) == 0u) { wlo[threadIdx.x >> 6] = lo; whi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = wlo[w] < lo ? wlo[w] : lo; hi = whi[w] > hi ? whi[w] : hi; }
        atomicMin(&lohi[0], lo);
        atomicMax(&lohi[1], hi);
    }
}

// bm[s][tt][w]: bit i of word w set <=> register 32 w + i of sketch s is <= lo + tt
__global__ void __launch_bounds__(256) hll_bitmaps_kernel(const uint8_t *__restrict__ img, uint32_t hdr, uint64_t stride, uint32_t m,
                                                          uint32_t lo, uint32_t band, uint32_t *__restrict__ bm)
{
    const uint32_t reg = blockIdx.x * 256u + threadIdx.x, s = blockIdx.y, words = m >> 5;
    const uint32_t v = img[(uint64_t)s * stride + hdr + reg];
    uint32_t *out = bm + (uint64_t)s * band * words + (reg >> 6) * 2u;
    for (uint32_t tt = 0; tt < band; ++tt) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(v <= lo + tt);
        if ((threadIdx.x & 63u) == 0u) { out[(uint64_t)tt * words] = (uint32_t)mask; out[(uint64_t)tt * words + 1u] = (uint32_t)(mask >> 32); }
    }
}

// WIDE: some register exceeds 32 (weights below 2^-32 need the second integer); ZERO: some register is 0 (the zero count is
// c_0).  Real sketches of genomes are neither: 48 accumulator registers less, twice the occupancy.
template <bool WIDE, bool ZERO>
__global__ void __launch_bounds__(256) hll_pairs_bitmap_kernel(const uint32_t *__restrict__ bmR, uint32_t n_ref, const uint32_t *__restrict__ bmQ,
                                                               uint32_t n_qry, uint32_t words, uint32_t lo, uint32_t band, uint32_t m,
                                                               uint32_t *__restrict__ out_zero, double *__restrict__ out_sum, int64_t tri)
{
    __shared__ __attribute__((aligned(16))) uint32_t R[HCW][HSTRIDE], Q[HCW][HSTRIDE];
    const uint32_t tid = threadIdx.x, tr = tid / 16, tq = tid % 16;
    const uint32_t r0 = blockIdx.y * HT, q0 = blockIdx.x * HT;
    if (tile_above_diagonal(tri, r0, HT, q0)) return;
    const uint32_t chunk = words < (uint32_t)HCW ? words : (uint32_t)HCW;       // m >= 1024: 32 words or more
    unsigned long long s1[HB][HB], s2[WIDE ? HB : 1][WIDE ? HB : 1];
    uint32_t zero[ZERO ? HB : 1][ZERO ? HB : 1];
#pragma unroll
    for (int i = 0; i < HB; ++i)
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            s1[i][j] = 0;
            if constexpr (WIDE) s2[i][j] = 0;
            if constexpr (ZERO) zero[i][j] = 0;
        }
    for (uint32_t tt = 0; tt < band; ++tt) {
        uint32_t cnt[HB][HB];
#pragma unroll
        for (int i = 0; i < HB; ++i)
#pragma unroll
            for (int j = 0; j < HB; ++j) cnt[i][j] = 0;
        for (uint32_t w0 = 0; w0 < words; w0 += chunk) {
            for (uint32_t i = tid; i < HT * chunk; i += 256) {
                const uint32_t row = i / chunk, col = i % chunk;
                R[col][row] = (r0 + row < n_ref) ? bmR[((uint64_t)(r0 + row) * band + tt) * words + w0 + col] : 0u;
                Q[col][row] = (q0 + row < n_qry) ? bmQ[((uint64_t)(q0 + row) * band + tt) * words + w0 + col] : 0u;
            }
            __syncthreads();
#pragma unroll 2
            for (uint32_t w = 0; w < chunk; ++w) {
                const uint4 av = *reinterpret_cast<const uint4 *>(&R[w][tr * HB]);
                const uint4 bv = *reinterpret_cast<const uint4 *>(&Q[w][tq * HB]);
                const uint32_t a[HB] = {av.x, av.y, av.z, av.w}, b[HB] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < HB; ++i)
#pragma unroll
                    for (int j = 0; j < HB; ++j) cnt[i][j] += (uint32_t)__popc(a[i] & b[j]);
            }
            __syncthreads();
        }
        const uint32_t e = lo + tt + 1u;                                        // c_t weighs 2^-(t+1); e <= 64 (host-checked)
#pragma unroll
        for (int i = 0; i < HB; ++i)
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                if constexpr (ZERO) { if (tt == 0u) zero[i][j] = cnt[i][j]; }           // (ZERO <=> lo == 0)
                if (!WIDE || e <= 32u) s1[i][j] += (unsigned long long)cnt[i][j] << (32u - e);
                else if constexpr (WIDE) s2[i][j] += (unsigned long long)cnt[i][j] << (64u - e);
            }
    }
    const uint32_t hi = lo + band;
#pragma unroll
    for (int i = 0; i < HB; ++i) {
        const uint32_t r = r0 + tr * HB + i;
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            const uint32_t q = q0 + tq * HB + j;
            if (r < n_ref && q < n_qry) {
                unsigned long long a1 = s1[i][j], a2 = 0;
                if constexpr (WIDE) a2 = s2[i][j];
                if (hi <= 32u) a1 += (unsigned long long)m << (32u - hi);
                else a2 += (unsigned long long)m << (64u - hi);
                const uint64_t o = (uint64_t)r * n_qry + q;
                uint32_t zc = 0;
                if constexpr (ZERO) zc = zero[i][j];
                out_zero[o] = zc;
                out_sum[o] = (double)a1 * 2.3283064365386963e-10 + (double)a2 * 5.421010862427522e-20;   // 2^-32, 2^-64
            }
        }
    }
}

// ---- UltraLogLog: distinct-count estimate of the union of every pair (utils.rs:260-270: UltraLogLog::merge + estimate) -----
// The union's registers are pack(unpack(a) | unpack(b)) (ull_merge_reg); both estimators of ultraloglog 0.1.6 (FGRA, ML)
// are functions of the register HISTOGRAM alone (ull_estimators.h), so no union sketch is materialised: a lane owns one
// (reference, query) pair and a private 256-bin histogram in LDS — bin-major, lane-minor, so that the 64 lanes of a wave
// always touch 64 different banks whatever values their registers hold — filled with fire-and-forget ds_add_u32.
// Narrow form (p <= 15: a bin counts at most 2^15): two 16-bit bins per word, 256 lanes = a 16 x 16 tile, 128 KiB;
// wide form (p >= 16): 32-bit bins, 128 lanes = an 8 x 16 tile, 128 KiB.  The estimator runs in the same kernel, one f64
// per pair leaves it.
constexpr int UQ = 16;                       // tile columns (queries)
constexpr int UCHUNK = 512;                  // registers per sketch per staging round
constexpr int UROW = UCHUNK / 4 + 1;

template <bool WIDE>
struct UllLaneHist {
    const uint32_t *w;
    uint32_t lanes, lane;
    __device__ __forceinline__ uint32_t operator()(uint32_t r) const
    {
        if constexpr (WIDE) return w[r * lanes + lane];
        else return (w[(r >> 1) * lanes + lane] >> (16u * (r & 1u))) & 0xFFFFu;
    }
};

template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? 128 : 256) ull_pairs_kernel(const uint8_t *__restrict__ ref, uint32_t n_ref,
                                                                      const uint8_t *__restrict__ qry, uint32_t n_qry, int p,
                                                                      uint32_t hdr, int estimator, double *__restrict__ out, int fixup,
                                                                      int64_t tri)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t LANES = WIDE ? 128u : 256u, TR = LANES / UQ, HW = (WIDE ? 256u : 128u) * LANES;
    if (tile_above_diagonal(tri, blockIdx.y * TR, TR, blockIdx.x * UQ)) return;
    // the histogram is addressed with raw LDS addresses (ds_add below): the dynamic array must start at LDS address 0, i.e. this
    // kernel must own no static LDS — which rules out __syncthreads_or() (HIP implements it with a __shared__ word)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds != 0u) __builtin_trap();   // ds_add below takes raw LDS addresses
    uint32_t *hist = lds;                                   // [bin (pair)][lane]
    if (fixup) {
        // second launch after ull_fgra_fast_kernel: only the tiles in which that kernel left a NaN (a pair that met an empty,
        // small-range or saturated register) are redone here; the others return at once
        const uint32_t fr = blockIdx.y * TR + threadIdx.x / UQ, fq = blockIdx.x * UQ + threadIdx.x % UQ;
        const bool mine = fr < n_ref && fq < n_qry && isnan(out[(uint64_t)fr * n_qry + fq]);
        if (threadIdx.x == 0) lds[0] = 0u;
        __syncthreads();
        if (mine) lds[0] = 1u;
        __syncthreads();
        const uint32_t any = lds[0];
        __syncthreads();
        if (!any) return;
    }
    uint32_t(*R)[UROW] = reinterpret_cast<uint32_t(*)[UROW]>(lds + HW);
    uint32_t(*Q)[UROW] = reinterpret_cast<uint32_t(*)[UROW]>(lds + HW + TR * UROW);
    const uint32_t tid = threadIdx.x, tr = tid / UQ, tq = tid % UQ;
    const uint32_t r0 = blockIdx.y * TR, q0 = blockIdx.x * UQ;
    const uint64_t m = 1ull << p, stride = (uint64_t)hdr + m;
    for (uint32_t i = tid; i < HW; i += LANES) hist[i] = 0;
    __syncthreads();
    for (uint64_t c0 = 0; c0 < m; c0 += UCHUNK) {
        const uint32_t n = m - c0 < (uint64_t)UCHUNK ? (uint32_t)(m - c0) : (uint32_t)UCHUNK;    // m >= 8: a multiple of 4
        for (uint32_t i = tid; i < (TR + UQ) * (n / 4); i += LANES) {
            const uint32_t row = i / (n / 4), col = i % (n / 4);
            const bool is_q = row >= TR;
            const uint32_t g = is_q ? q0 + row - TR : r0 + row;
            uint32_t v = 0;
            if (g < (is_q ? n_qry : n_ref)) {
                const uint8_t *s = (is_q ? qry : ref) + (uint64_t)g * stride + hdr + c0 + 4 * col;
                v = s[0] | (s[1] << 8) | (s[2] << 16) | ((uint32_t)s[3] << 24);
            }
            if (is_q) Q[row - TR][col] = v; else R[row][col] = v;
        }
        __syncthreads();
        for (uint32_t w = 0; w < n / 4; ++w) {
            const uint32_t a = R[tr][w], b = Q[tq][w];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t r = ull_merge_fast((a >> (8 * j)) & 0xFFu, (b >> (8 * j)) & 0xFFu);
                if constexpr (WIDE) asm volatile("ds_add_u32 %0, %1" ::"v"((r * LANES + tid) << 2), "v"(1u) : "memory");
                else asm volatile("ds_add_u32 %0, %1" ::"v"(((r >> 1) * LANES + tid) << 2), "v"(1u << (16u * (r & 1u))) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (r0 + tr < n_ref && q0 + tq < n_qry) {
        const UllLaneHist<WIDE> h{hist, LANES, tid};
        double *dst = out + (uint64_t)(r0 + tr) * n_qry + q0 + tq;
        if (!fixup || isnan(*dst)) *dst = estimator == 1 ? ull::ml(h, p) : ull::fgra(h, p);
    }
}

// FGRA when every merged register is "regular" (largest update value >= 3, not saturated — every register of a genome-sized
// sketch): n = lambda_p * (sum_i g(r_i))^(-1/tau) needs no histogram.  A lane owns one pair and adds g(r) from a 256-entry f64
// table in LDS; the entries of the special registers (empty, 4p-4, 4p, 4p+2, >= 252) are NaN, so a pair that meets one
// comes out NaN and the histogram kernel redoes it (fixup launch).  2 KiB + 33 KiB of LDS instead of 128 KiB: full occupancy.
constexpr int FCHUNK = 1024;
constexpr int FROW = FCHUNK / 4 + 1;

__global__ void __launch_bounds__(256) ull_fgra_fast_kernel(const uint8_t *__restrict__ ref, uint32_t n_ref,
                                                            const uint8_t *__restrict__ qry, uint32_t n_qry, int p, uint32_t hdr,
                                                            double *__restrict__ out, int64_t tri)
{
    __shared__ double G[256];
    __shared__ uint32_t R[UQ][FROW], Q[UQ][FROW];
    const uint32_t tid = threadIdx.x, tr = tid / UQ, tq = tid % UQ;
    const uint32_t r0 = blockIdx.y * UQ, q0 = blockIdx.x * UQ;
    if (tile_above_diagonal(tri, r0, UQ, q0)) return;
    const uint64_t m = 1ull << p, stride = (uint64_t)hdr + m;
    {
        const uint32_t r = tid, off = 4u * (uint32_t)p + 4u;
        G[r] = (r >= off && r < 252u) ? ull::eta(r & 3u) * pow(2.0, -ull::TAU * (double)((r >> 2) - (uint32_t)p + 2u)) : __longlong_as_double(0x7FF8000000000000ll);
    }
    __syncthreads();
    double sum = 0.0;
    for (uint64_t c0 = 0; c0 < m; c0 += FCHUNK) {
        const uint32_t n = m - c0 < (uint64_t)FCHUNK ? (uint32_t)(m - c0) : (uint32_t)FCHUNK;
        for (uint32_t i = tid; i < 2 * UQ * (n / 4); i += 256) {
            const uint32_t row = i / (n / 4), col = i % (n / 4);
            const bool is_q = row >= UQ;
            const uint32_t g = is_q ? q0 + row - UQ : r0 + row;
            uint32_t v = 0;
            if (g < (is_q ? n_qry : n_ref)) {
                const uint8_t *s = (is_q ? qry : ref) + (uint64_t)g * stride + hdr + c0 + 4 * col;
                v = s[0] | (s[1] << 8) | (s[2] << 16) | ((uint32_t)s[3] << 24);
            }
            if (is_q) Q[row - UQ][col] = v; else R[row][col] = v;
        }
        __syncthreads();
        for (uint32_t w = 0; w < n / 4; ++w) {
            const uint32_t a = R[tr][w], b = Q[tq][w];
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += G[ull_merge_fast((a >> (8 * j)) & 0xFFu, (b >> (8 * j)) & 0xFFu)];
        }
        __syncthreads();
    }
    if (r0 + tr < n_ref && q0 + tq < n_qry) {
        const double factor = pow((double)m, 1.0 + 1.0 / ull::TAU) / (1.0 + ull::V * (1.0 + ull::TAU) / (2.0 * (double)m));
        out[(uint64_t)(r0 + tr) * n_qry + q0 + tq] = isnan(sum) ? sum : factor * pow(sum, -1.0 / ull::TAU);
    }
}

hipError_t launch_ull_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, int p, uint32_t hdr,
                            int estimator, double *d_est, hipStream_t stream, int64_t tri)
{
    if (n_ref == 0 || n_qry == 0) return hipSuccess;
    const bool wide = p >= 16;
    const uint32_t lanes = wide ? 128u : 256u, tr = lanes / UQ;
    const size_t lds = ((wide ? 256u : 128u) * lanes + (tr + UQ) * UROW) * 4u;
    dim3 grid((n_qry + UQ - 1) / UQ, (n_ref + tr - 1) / tr);
    hipError_t e;
    int fixup = 0;
    static const bool no_fast = getenv("LASH_ULL_NO_FAST") != nullptr;    // A/B knob (tools/dist_rate.py)
    if (estimator == 0 && !no_fast) {                     // FGRA: the histogram-free kernel first, the histogram kernel only where it gave up
        hipLaunchKernelGGL(ull_fgra_fast_kernel, dim3((n_qry + UQ - 1) / UQ, (n_ref + UQ - 1) / UQ), dim3(256), 0, stream, d_ref, n_ref,
                           d_qry, n_qry, p, hdr, d_est, tri);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        fixup = 1;
    }
    if (wide) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(ull_pairs_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(ull_pairs_kernel<true>, grid, dim3(lanes), lds, stream, d_ref, n_ref, d_qry, n_qry, p, hdr, estimator, d_est, fixup, tri);
    } else {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(ull_pairs_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(ull_pairs_kernel<false>, grid, dim3(lanes), lds, stream, d_ref, n_ref, d_qry, n_qry, p, hdr, estimator, d_est, fixup, tri);
    }
    return hipGetLastError();
}

hipError_t launch_hll_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, int p, uint32_t hdr,
                            uint32_t *d_zero, double *d_sum, hipStream_t stream, int64_t tri)
{
    if (n_ref == 0 || n_qry == 0) return hipSuccess;
    dim3 grid((n_qry + DT - 1) / DT, (n_ref + DT - 1) / DT);
    hipLaunchKernelGGL(hll_pairs_kernel, grid, dim3(256), 0, stream, d_ref, n_ref, d_qry, n_qry, p, hdr, d_zero, d_sum, tri);
    return hipGetLastError();
}

hipError_t launch_hmh_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, uint32_t hdr,
                            uint64_t stride, uint32_t *d_c, uint32_t *d_n, hipStream_t stream, int64_t tri)
{
    if (n_ref == 0 || n_qry == 0) return hipSuccess;
    dim3 grid((n_qry + HT - 1) / HT, (n_ref + HT - 1) / HT);
    hipLaunchKernelGGL(hmh_pairs_kernel, grid, dim3(256), 0, stream, d_ref, n_ref, d_qry, n_qry, hdr, stride, d_c, d_n, tri);
    return hipGetLastError();
}

hipError_t launch_hll_minmax(const uint8_t *d_img, uint32_t n, int p, uint32_t hdr, uint32_t *d_lohi, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint32_t m = 1u << p;                                          // p >= 10: a multiple of 1 024
    hipLaunchKernelGGL(hll_minmax_kernel, dim3(n), dim3(256), 0, stream, d_img, hdr, (uint64_t)hdr + m, m, d_lohi);
    return hipGetLastError();
}

hipError_t launch_hll_bitmaps(const uint8_t *d_img, uint32_t n, int p, uint32_t hdr, uint32_t lo, uint32_t band, uint32_t *d_bm, hipStream_t stream)
{
    if (n == 0 || band == 0) return hipSuccess;
    const uint32_t m = 1u << p;
    for (uint32_t s0 = 0; s0 < n; s0 += 65535u) {                       // grid.y limit
        const uint32_t ns = std::min(65535u, n - s0);
        hipLaunchKernelGGL(hll_bitmaps_kernel, dim3(m / 256, ns), dim3(256), 0, stream, d_img + (uint64_t)s0 * ((uint64_t)hdr + m), hdr,
                           (uint64_t)hdr + m, m, lo, band, d_bm + (uint64_t)s0 * band * (m >> 5));
    }
    return hipGetLastError();
}

hipError_t launch_hll_pairs_bitmap(const uint32_t *d_bm_ref, uint32_t n_ref, const uint32_t *d_bm_qry, uint32_t n_qry, int p, uint32_t lo,
                                   uint32_t band, uint32_t *d_zero, double *d_sum, hipStream_t stream, int64_t tri)
{
    if (n_ref == 0 || n_qry == 0) return hipSuccess;
    dim3 grid((n_qry + HT - 1) / HT, (n_ref + HT - 1) / HT);
    const bool wide = lo + band > 32u, zero = lo == 0u;
    auto kern = wide ? (zero ? hll_pairs_bitmap_kernel<true, true> : hll_pairs_bitmap_kernel<true, false>)
                     : (zero ? hll_pairs_bitmap_kernel<false, true> : hll_pairs_bitmap_kernel<false, false>);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, d_bm_ref, n_ref, d_bm_qry, n_qry, (1u << p) >> 5, lo, band, 1u << p, d_zero, d_sum, tri);
    return hipGetLastError();
}

hipError_t launch_collision_vectors(const double *d_card, uint32_t n, double *d_P, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(collision_vectors_kernel, dim3(EC_CELLS / 256, n), dim3(256), 0, stream, d_card, d_P);
    return hipGetLastError();
}

hipError_t launch_collision_gemm(const double *d_A, uint32_t m, const double *d_B, uint32_t n, double *d_X, hipStream_t stream)
{
    if (m == 0 || n == 0) return hipSuccess;
    const uint32_t tx = (n + EC_T - 1) / EC_T, ty = (m + EC_T - 1) / EC_T, tiles = tx * ty;
    hipLaunchKernelGGL(collision_gemm_kernel, dim3(((tiles + 7u) / 8u) * 8u), dim3(256), 0, stream, d_A, m, d_B, n, d_X, tx, tiles);
    return hipGetLastError();
}

}  // namespace lash
