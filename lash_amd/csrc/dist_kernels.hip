// dist_kernels.hip — pair statistics for `lash dist` on HyperMinHash and HyperLogLog sketches.
//
// Replaces the register scan inside hyperminhash's Sketch::similarity, called once per (reference, query) pair from
// /root/reference/src/utils.rs:164:   C = #{i : a_i != 0 and a_i == b_i},   N = #{i : a_i != 0 or b_i != 0}.
// The O(N_ref * N_qry * 16384) part runs here; cardinalities and the collision correction are O(pairs) host work.
//
// One 256-thread workgroup owns a 16 x 16 tile of pairs and walks the 16 384 registers in chunks of 512 staged in
// LDS (rows padded to 257 words so that the 16 query rows of a column land on different banks; the reference row is
// a broadcast).  Two u16 registers per u32 are compared with SWAR zero-half tests.
#include <hip/hip_runtime.h>

#include "lash_kernels.h"

namespace lash {

constexpr int DT = 16;                 // tile edge (pairs)
constexpr int DCHUNK_WORDS = 256;      // 512 registers per sketch per chunk
constexpr int DROW = DCHUNK_WORDS + 1; // padded LDS row

__device__ __forceinline__ uint32_t zero_halves(uint32_t x)
{
    // 0x8000 in every 16-bit half of x that is zero (exact, no carries across halves)
    return ~(((x & 0x7FFF7FFFu) + 0x7FFF7FFFu) | x | 0x7FFF7FFFu);
}

__global__ void __launch_bounds__(256) hmh_pairs_kernel(const uint32_t *__restrict__ ref, uint32_t n_ref,
                                                        const uint32_t *__restrict__ qry, uint32_t n_qry,
                                                        uint32_t *__restrict__ out_c, uint32_t *__restrict__ out_n)
{
    __shared__ uint32_t R[DT][DROW], Q[DT][DROW];
    const uint32_t tid = threadIdx.x, tr = tid / DT, tq = tid % DT;
    const uint32_t r0 = blockIdx.y * DT, q0 = blockIdx.x * DT;
    constexpr uint32_t WORDS = HMH_M / 2;                       // 8192 u32 per sketch
    uint32_t c = 0, n = 0;
    for (uint32_t w0 = 0; w0 < WORDS; w0 += DCHUNK_WORDS) {
        for (uint32_t i = tid; i < DT * DCHUNK_WORDS; i += 256) {
            const uint32_t row = i / DCHUNK_WORDS, col = i % DCHUNK_WORDS;
            R[row][col] = (r0 + row < n_ref) ? ref[(uint64_t)(r0 + row) * WORDS + w0 + col] : 0u;
            Q[row][col] = (q0 + row < n_qry) ? qry[(uint64_t)(q0 + row) * WORDS + w0 + col] : 0u;
        }
        __syncthreads();
#pragma unroll 8
        for (uint32_t w = 0; w < DCHUNK_WORDS; ++w) {
            const uint32_t a = R[tr][w], b = Q[tq][w];
            const uint32_t za = zero_halves(a), zb = zero_halves(b);
            c += (uint32_t)__builtin_popcount(zero_halves(a ^ b) & ~za);          // equal and non-zero
            n += 2u - (uint32_t)__builtin_popcount(za & zb);                      // not both zero
        }
        __syncthreads();
    }
    if (r0 + tr < n_ref && q0 + tq < n_qry) {
        out_c[(uint64_t)(r0 + tr) * n_qry + q0 + tq] = c;
        out_n[(uint64_t)(r0 + tr) * n_qry + q0 + tq] = n;
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
                                                        const uint8_t *__restrict__ qry, uint32_t n_qry, int p,
                                                        uint32_t *__restrict__ out_zero, double *__restrict__ out_sum)
{
    __shared__ uint32_t R[DT][HROW], Q[DT][HROW];
    const uint32_t tid = threadIdx.x, tr = tid / DT, tq = tid % DT;
    const uint32_t r0 = blockIdx.y * DT, q0 = blockIdx.x * DT;
    const uint32_t m = 1u << p;
    const uint64_t stride = 33ull + m;
    unsigned long long s1 = 0, s2 = 0;          // sum of 2^(32-r) over r <= 32 ; sum of 2^(64-r) over r > 32
    uint32_t zero = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += HCHUNK) {
        const uint32_t n = m - c0 < (uint32_t)HCHUNK ? m - c0 : (uint32_t)HCHUNK;     // m >= 16: a multiple of 4
        for (uint32_t i = tid; i < DT * (n / 4); i += 256) {
            const uint32_t row = i / (n / 4), col = i % (n / 4);
            uint32_t a = 0, b = 0;
            if (r0 + row < n_ref) {
                const uint8_t *s = ref + (uint64_t)(r0 + row) * stride + 33 + c0 + 4 * col;
                a = s[0] | (s[1] << 8) | (s[2] << 16) | ((uint32_t)s[3] << 24);
            }
            if (q0 + row < n_qry) {
                const uint8_t *s = qry + (uint64_t)(q0 + row) * stride + 33 + c0 + 4 * col;
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

hipError_t launch_hll_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, int p,
                            uint32_t *d_zero, double *d_sum, hipStream_t stream)
{
    if (n_ref == 0 || n_qry == 0) return hipSuccess;
    dim3 grid((n_qry + DT - 1) / DT, (n_ref + DT - 1) / DT);
    hipLaunchKernelGGL(hll_pairs_kernel, grid, dim3(256), 0, stream, d_ref, n_ref, d_qry, n_qry, p, d_zero, d_sum);
    return hipGetLastError();
}

hipError_t launch_hmh_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, uint32_t *d_c,
                            uint32_t *d_n, hipStream_t stream)
{
    if (n_ref == 0 || n_qry == 0) return hipSuccess;
    dim3 grid((n_qry + DT - 1) / DT, (n_ref + DT - 1) / DT);
    hipLaunchKernelGGL(hmh_pairs_kernel, grid, dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(d_ref), n_ref,
                       reinterpret_cast<const uint32_t *>(d_qry), n_qry, d_c, d_n);
    return hipGetLastError();
}

}  // namespace lash
