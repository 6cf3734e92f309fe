// dist_kernels.hip — pair statistics for `lash dist` on HyperMinHash sketches.
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
