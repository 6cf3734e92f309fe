// pair_planes.hip — HyperMinHash pair counts through register BIT PLANES (BASELINE configs[3]).
//
// Replaces the register scan inside hyperminhash's Sketch::similarity, called once per (reference, query) pair from
// /root/reference/src/utils.rs:164:   C = #{i : a_i != 0 and a_i == b_i},   N = #{i : a_i != 0 or b_i != 0}
// for collections: all-vs-all on 10^5 sketches is 5 * 10^9 pairs x 16 384 registers.
//
// Formulation.  A sketch's 16 384 u16 registers are transposed ONCE into 16 bit planes of 512 words (plane b, word w, bit i =
// bit b of register 32 w + i) plus the plane of non-zero registers.  For one word of one pair
//     diff = OR_b (A_b ^ B_b)            16 instructions: v_xor, then 15 x v_bitop3 (d | (a ^ b))
//     C   += popcount(nzA & ~diff)       v_bfi / v_bitop3 + v_bcnt (accumulating)
//     N   += popcount(nzA | nzB)         v_or + v_bcnt
// = 20 VALU instructions per 32 registers (0.63 per register pair; the u16-pair kernel of dist_kernels.hip needs 2.75).  When
// every register of both sketches is non-zero — any genome with far more than 16 384 distinct k-mers — N = 16 384 and
// C = 16 384 - sum popcount(diff): 17 instructions per 32 registers.
//
// Measured instruction costs on gfx950 (tools/ubench_ops.hip, 4 waves per SIMD): a VOP2 on two VGPRs issues in 2.9 cycles per
// wave, everything else used here — v_bitop3, v_bcnt, any form with an SGPR source — in 4.4.  17 x 4.4 = 75 cycles per word pair
// and wave bound the kernel at ~4.3e9 pairs/s (tools/ubench_planes.hip, EXP=2: this kernel's arithmetic with the operand traffic
// removed); xor + or as two VOP2 would be 5.8 cycles per plane instead of 4.4.
//
// Mapping (VALU-issue bound; everything else is arranged to stay out of the way):
//   * a LANE owns TWO reference sketches (rows): their plane words come from the set's row layout T[word][plane][sketch], so a
//     wave's 2 x 64 rows are coalesced 256-byte loads, 32 per word, prefetched one word ahead;
//   * the query sketch (column) is WAVE-UNIFORM: the workgroup stages the tile's plane words of the next word in LDS (one
//     coalesced 4-byte load per thread from the column layout S[word][sketch][plane], two words ahead), every lane reads them
//     back with broadcast ds_read_b128 (conflict-free; 16 LDS cycles per column and wave against 150 of arithmetic).
//     [First version: the column words as scalar operands straight from s_load_dwordx16.  Scalar loads return out of order, so
//     lgkmcnt(0) is the only wait there is; with the SGPR file holding two column pairs the loads had 34 instructions of cover:
//     2.5e9 pairs/s, 72 % of the wave cycles in s_waitcnt, and still only 3.4e9 with every load hitting one cache line.]
//   * a wave sweeps a tile of QT columns per word with one accumulator register per (row, column).  A 256-thread workgroup =
//     4 waves = 512 rows sharing the column tile;
//   * tiles that lie wholly above the diagonal of a same-set (triangular, utils.rs:158-160) call return at once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>

#include "lash_device.h"
#include "lash_kernels.h"

#ifndef LASH_PLANES_QT_FULL
#define LASH_PLANES_QT_FULL 32
#endif
#ifndef LASH_PLANES_QT_GEN
#define LASH_PLANES_QT_GEN 16
#endif

namespace lash {

constexpr uint32_t PL_WORDS = HMH_M / 32;      // 512 words per plane
constexpr uint32_t PL_N = 16;                  // bit planes of a u16 register
constexpr uint32_t PL_TN = 17;                 // planes per word in the row layout: 16 + the plane of non-zero registers
constexpr uint32_t PL_SN = 20;                 // words per (word, member) in the column layout: 16 + non-zero plane + 3 of padding (16-byte reads)
constexpr uint32_t PL_ROWS = 512;              // rows of a pair-kernel workgroup (row pitch of T is a multiple of this)
constexpr uint32_t PL_COLS = 64;               // S is padded to a multiple of this many members

// ---- planes: row layout T[w][17][ld]; column layout S[w][n_pad][20]; nzcount[s] ---------------------------------------------
// One workgroup = 64 consecutive set members x one pair of plane words (64 registers).  A wave turns a member's 64 registers
// into 17 ballots; the 2 x 17 plane words of the 64 members meet in LDS and leave as contiguous runs in both layouts.
__global__ void __launch_bounds__(256) hmh_planes_kernel(const uint8_t *__restrict__ img, uint32_t hdr, uint64_t stride, uint32_t n,
                                                         uint32_t *__restrict__ T, uint32_t ldT, uint32_t *__restrict__ S, uint32_t n_pad,
                                                         uint32_t *__restrict__ nzcount)
{
    __shared__ uint32_t tile[2][PL_SN][64];
    const uint32_t s0 = blockIdx.x * 64u, wp = blockIdx.y, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 2u * 3u * 64u; i += 256u) tile[i / 192u][PL_TN + (i / 64u) % 3u][i & 63u] = 0u;   // the padding words
    for (uint32_t j = wave; j < 64u; j += 4u) {
        const uint32_t s = s0 + j;
        uint32_t reg = 0;
        if (s < n) {
            const uint8_t *r = img + (uint64_t)s * stride + hdr + 2ull * (64u * wp + lane);   // (byte order is irrelevant to == and != 0)
            reg = (uint32_t)r[0] | ((uint32_t)r[1] << 8);
        }
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t b = 0; b < PL_N; ++b) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64((reg >> b) & 1u);
            const uint32_t v = (lane & 16u) ? (uint32_t)(m >> 32) : (uint32_t)m;
            mine = (lane & 15u) == b ? v : mine;
        }
        if (lane < 32u) tile[lane >> 4][lane & 15u][j] = mine;               // lanes 0..15: word 2 wp; 16..31: word 2 wp + 1
        const unsigned long long z = __builtin_amdgcn_ballot_w64(reg != 0u);
        if (lane < 2u) tile[lane][PL_N][j] = lane ? (uint32_t)(z >> 32) : (uint32_t)z;
        if (lane == 0u && s < n && nzcount) atomicAdd(&nzcount[s], (uint32_t)__builtin_popcountll(z));
    }
    __syncthreads();
    if (T)                                                                     // for each (word, plane) the 64 members are consecutive
        for (uint32_t i = threadIdx.x; i < 2u * PL_TN * 64u; i += 256u) {
            const uint32_t j = i & 63u, b = (i >> 6) % PL_TN, h = (i >> 6) / PL_TN;
            if (s0 + j < ldT) T[(uint64_t)((2u * wp + h) * PL_TN + b) * ldT + s0 + j] = tile[h][b][j];
        }
    if (S)                                                                     // for each word: member-major, 20 words each
        for (uint32_t i = threadIdx.x; i < 2u * 64u * PL_SN; i += 256u) {
            const uint32_t b = i % PL_SN, j = (i / PL_SN) & 63u, h = i / (PL_SN * 64u);
            if (s0 + j < n_pad) S[((uint64_t)(2u * wp + h) * n_pad + s0 + j) * PL_SN + b] = tile[h][b][j];
        }
}

// ---- the pair kernel ---------------------------------------------------------------------------------------------------------
struct PlanePairArgs {
    const uint32_t *T;        // rows:    T[(w * 17 + b) * ldT + row]        (b = 16: the non-zero plane)
    const uint32_t *S;        // columns: S[(w * n_pad + col) * 20 + b]      (padded with zero sketches to a multiple of PL_COLS)
    uint32_t ldT;             // row pitch of T (a multiple of PL_ROWS, zero padded)
    uint32_t n_pad;           // column pitch of S
    uint32_t row0, n_rows;    // rows [row0, row0 + n_rows) of the row set
    uint32_t n_cols;          // columns [0, n_cols) of the column set
    int32_t  triangle;        // != 0: row set == column set; only pairs with col <= row are wanted
    uint32_t *out_c, *out_n;  // [n_rows][ld_out]
    uint64_t ld_out;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool FULL, int QT>
__global__ void __launch_bounds__(256, 2) hmh_pairs_planes_kernel(PlanePairArgs a)
{
    constexpr int SLICE = QT * PL_SN;                                          // words of one column slice
    constexpr int SL_PER_T = (SLICE + 255) / 256;
    constexpr uint32_t NB = FULL ? PL_N : PL_TN;                               // planes the arithmetic reads
    __shared__ __attribute__((aligned(16))) uint32_t sb[3][SL_PER_T * 256];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (tell hipcc it is wave-uniform:
    // otherwise every buffer load below becomes a waterfall loop over "divergent" resource descriptors)
    const uint32_t rloc = blockIdx.y * PL_ROWS + wave * 128u + lane;          // first of the lane's two rows (the other: + 64)
    const uint32_t col0 = blockIdx.x * (uint32_t)QT;
    // triangular call: the workgroup's last row is row0 + blockIdx.y * 512 + 511; columns beyond it are never printed
    if (a.triangle && col0 > a.row0 + blockIdx.y * PL_ROWS + (PL_ROWS - 1u)) return;
    // addresses: wave-uniform base (SGPRs) + the lane's 32-bit offset, so that no 64-bit per-lane address is ever formed
    // (with per-lane pointers hipcc precomputed one register pair per plane: 64 registers of addresses, spills)
    const uint32_t *__restrict__ T = a.T + a.row0 + blockIdx.y * PL_ROWS + wave * 128u;     // (T is padded: always readable)
    const uint32_t *__restrict__ S = a.S + (uint64_t)col0 * PL_SN;
    const uint64_t ld = a.ldT, sp = (uint64_t)a.n_pad * PL_SN;

    uint32_t acc_c[2][QT], acc_n[FULL ? 1 : 2][FULL ? 1 : QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        acc_c[0][q] = acc_c[1][q] = 0;
        if constexpr (!FULL) acc_n[0][q] = acc_n[1][q] = 0;
    }

    struct Rows { uint32_t p[2][NB]; };
    // buffer loads: one resource per word (wave-uniform base), the plane as scalar offset, the lane as the only vector offset
    const uint32_t lane4 = lane * 4u;
    auto load_rows = [&](uint32_t w, Rows &r) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(T + (uint64_t)w * PL_TN * ld), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (uint32_t b = 0; b < NB; ++b) {
            const uint32_t so = b * (uint32_t)ld * 4u;                                  // < 2^31: ld < 2^24
            r.p[0][b] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4, so, 0);
            r.p[1][b] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4 + 256u, so, 0);
        }
    };
    // the tile's column slice of word w: QT * 20 contiguous words of S, staged in whole 256-word rounds (the words past the
    // slice belong to the next columns — S has 1 024 words of slack at its end — and are never read back: no conditional code
    // in the loop, which sends hipcc's register allocation astray)
    uint32_t st[SL_PER_T];
    auto fetch_slice = [&](uint32_t w) {
        const uint32_t *__restrict__ src = S + (uint64_t)w * sp;                       // uniform
#pragma unroll
        for (int i = 0; i < SL_PER_T; ++i) st[i] = src[threadIdx.x + 256u * i];
    };
    auto put_slice = [&](uint32_t buf) {
#pragma unroll
        for (int i = 0; i < SL_PER_T; ++i) sb[buf][threadIdx.x + 256u * i] = st[i];
    };
    // one word of the lane's two rows against the tile's QT columns.  The column's words are read one column ahead (hipcc
    // on its own issues the reads and waits for them at once: the LDS latency of every column exposed), and
    // sched_barrier keeps it from hoisting ALL of a sweep's reads to the top (256 registers of operands, spills).
    constexpr int NV = FULL ? 4 : 5;                                            // 16-byte reads per column
    auto read_col = [&](uint32_t buf, int q, uint32_t (&bw)[4 * NV]) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {                                          // wave-uniform address: broadcast reads
            const u32x4 x = *reinterpret_cast<const u32x4 *>(&sb[buf][q * PL_SN + 4 * v]);
            bw[4 * v] = x.x; bw[4 * v + 1] = x.y; bw[4 * v + 2] = x.z; bw[4 * v + 3] = x.w;
        }
    };
    auto column = [&](int q, const Rows &r, const uint32_t (&bw)[4 * NV]) {
        uint32_t d0 = r.p[0][0] ^ bw[0], d1 = r.p[1][0] ^ bw[0];
#pragma unroll
        for (uint32_t b = 1; b < PL_N; ++b) {
            d0 = __builtin_amdgcn_bitop3_b32(r.p[0][b], d0, bw[b], 0xde);       // d | (a ^ b), one instruction
            d1 = __builtin_amdgcn_bitop3_b32(r.p[1][b], d1, bw[b], 0xde);
        }
        if constexpr (FULL) {
            acc_c[0][q] += (uint32_t)__builtin_popcount(d0);                  // registers that differ
            acc_c[1][q] += (uint32_t)__builtin_popcount(d1);
        } else {
            acc_c[0][q] += (uint32_t)__builtin_popcount(r.p[0][PL_N] & ~d0);
            acc_c[1][q] += (uint32_t)__builtin_popcount(r.p[1][PL_N] & ~d1);
            acc_n[0][q] += (uint32_t)__builtin_popcount(r.p[0][PL_N] | bw[PL_N]);
            acc_n[1][q] += (uint32_t)__builtin_popcount(r.p[1][PL_N] | bw[PL_N]);
        }
    };
    auto sweep = [&](uint32_t buf, const Rows &r) {
        uint32_t b0[4 * NV], b1[4 * NV];
        read_col(buf, 0, b0);
#pragma unroll
        for (int q = 0; q < QT; q += 2) {
            read_col(buf, q + 1, b1);
            __builtin_amdgcn_sched_barrier(0);
            column(q, r, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 2 < QT) read_col(buf, q + 2, b0);
            __builtin_amdgcn_sched_barrier(0);
            column(q + 1, r, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // pipeline: rows of word w+1 in flight while word w is swept; column slices two words ahead in registers, one ahead in LDS.
    // One barrier per word: the slice written during word w goes to buffer (w+1) % 3, last read during word w-2.
    Rows r0, r1;
    load_rows(0, r0);
    fetch_slice(0);
    put_slice(0);
    fetch_slice(1);
#pragma unroll 1
    for (uint32_t w = 0; w < PL_WORDS; w += 2) {
        __syncthreads();
        put_slice((w + 1) % 3);
        fetch_slice(w + 2 < PL_WORDS ? w + 2 : w);
        load_rows(w + 1, r1);
        sweep(w % 3, r0);
        __syncthreads();
        put_slice((w + 2) % 3);
        fetch_slice(w + 3 < PL_WORDS ? w + 3 : w);
        load_rows(w + 2 < PL_WORDS ? w + 2 : w, r0);                          // (past the end: a harmless reload)
        sweep((w + 1) % 3, r1);
    }

#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t rr = rloc + 64u * h;
        if (rr < a.n_rows) {
            uint32_t *oc = a.out_c + (uint64_t)rr * a.ld_out + col0;
            uint32_t *on = a.out_n + (uint64_t)rr * a.ld_out + col0;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                if (col0 + (uint32_t)q < a.n_cols) {
                    oc[q] = FULL ? HMH_M - acc_c[h][q] : acc_c[h][q];
                    on[q] = FULL ? HMH_M : acc_n[FULL ? 0 : h][FULL ? 0 : q];
                }
            }
        }
    }
}

hipError_t launch_hmh_planes(const uint8_t *d_img, uint32_t hdr, uint64_t stride, uint32_t n, uint32_t *d_T, uint32_t ldT, uint32_t *d_S,
                             uint32_t n_pad, uint32_t *d_nzcount, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint32_t span = std::max(std::max(d_T ? ldT : 0u, d_S ? n_pad : 0u), n);
    hipLaunchKernelGGL(hmh_planes_kernel, dim3((span + 63u) / 64u, PL_WORDS / 2), dim3(256), 0, stream, d_img, hdr, stride, n, d_T, ldT, d_S, n_pad,
                       d_nzcount);
    return hipGetLastError();
}

uint32_t hmh_planes_col_pad() { return PL_COLS; }
uint32_t hmh_planes_row_pad() { return PL_ROWS; }
size_t   hmh_planes_T_words(uint32_t ldT) { return (size_t)PL_WORDS * PL_TN * ldT; }
size_t   hmh_planes_S_words(uint32_t n_pad) { return (size_t)PL_WORDS * PL_SN * n_pad + 1024; }   // + the slack fetch_slice may read

hipError_t launch_hmh_pairs_planes(const uint32_t *d_T, uint32_t ldT, uint32_t row0, uint32_t n_rows, const uint32_t *d_S, uint32_t n_pad,
                                   uint32_t n_cols, bool full, bool triangle, uint32_t *d_c, uint32_t *d_n, uint64_t ld_out, hipStream_t stream)
{
    if (n_rows == 0 || n_cols == 0) return hipSuccess;
    PlanePairArgs a{d_T, d_S, ldT, n_pad, row0, n_rows, n_cols, triangle ? 1 : 0, d_c, d_n, ld_out};
    constexpr int QF = LASH_PLANES_QT_FULL, QG = LASH_PLANES_QT_GEN;
    if (full) {
        dim3 grid((n_cols + QF - 1) / QF, (n_rows + PL_ROWS - 1) / PL_ROWS);
        hipLaunchKernelGGL((hmh_pairs_planes_kernel<true, QF>), grid, dim3(256), 0, stream, a);
    } else {
        dim3 grid((n_cols + QG - 1) / QG, (n_rows + PL_ROWS - 1) / PL_ROWS);
        hipLaunchKernelGGL((hmh_pairs_planes_kernel<false, QG>), grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace lash
