// ubench_isa.hip — issue cost of the instruction FORMS the sketch kernels' hot loops are made of, gfx950, at the kernels' own
// occupancy (4 waves per SIMD): one row per (opcode, operand class), cycles per wave-instruction per SIMD.  tools/isa_cost.py
// prices a compiler listing with this table (profiles/r04/isa_cost/costs.json, written by `ubench_isa --json FILE`).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_isa tools/ubench_isa.hip
//
// Every row runs 16 instructions per loop iteration on 8 independent accumulators (dependent distance 8 instructions), 4
// workgroups of 4 waves per CU; the loop overhead (s_add, s_cmp, s_cbranch) is 3 scalar instructions per 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Row { const char *name; const char *cls; double ns, ghz; int per_iter; };
static std::vector<Row> g_rows;

#define R8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#define Q8(S) S(q0) S(q1) S(q2) S(q3) S(q4) S(q5) S(q6) S(q7)

template <int OP>
__global__ void __launch_bounds__(256) bench(uint32_t *sink, int iters, uint32_t sarg, uint64_t sarg64, unsigned long long *clk)
{
    extern __shared__ uint32_t lds[];
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9E3779B9u, a2 = a0 * 3 + 7, a3 = a1 * 5 + 11;
    uint32_t a4 = a0 + 0x1234567, a5 = a1 + 0x7654321, a6 = a2 ^ 0xdeadbeef, a7 = a3 ^ 0xcafebabe;
    uint64_t q0 = a0 | ((uint64_t)a1 << 32), q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 9, q5 = q0 * 11, q6 = q0 * 13, q7 = q0 * 15;
    const uint32_t c = 0x85EBCA97u + threadIdx.x, d = 0x9E3779F9u ^ threadIdx.x;   // VGPR-resident "constants"
    const uint64_t cq = ((uint64_t)d << 32) | c;
    uint32_t s = sarg;                                   // wave-uniform -> SGPR
    uint64_t sq = sarg64;
    (void)sq; (void)cq;
    for (int i = 0; i < iters; ++i) {
        // ---- simple VOP2 ops by operand class ----
        if constexpr (OP == 0) {
#define S(a) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 1) {
#define S(a) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 2) {
#define S(a) asm volatile("v_xor_b32 %0, 5, %0" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 3) {
#define S(a) asm volatile("v_xor_b32 %0, 0x9e3779b1, %0" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 4) {
#define S(a) asm volatile("v_lshrrev_b32 %0, 5, %0\n\tv_add_u32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S)
#undef S
        } else if constexpr (OP == 5) {
#define S(a) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 6) {
#define S(a) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 7) {
#define S(a) asm volatile("v_add_u32 %0, 8, %0" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 8) {
#define S(a) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 9) {
#define S(a) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 10) {
#define S(a) asm volatile("v_and_b32 %0, 0x3ffff, %0" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 11) {
#define S(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c) : );
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(c) : "vcc");
            R8(S) R8(S)
#undef S
        // ---- VOP3 forms ----
        } else if constexpr (OP == 12) {
#define S(a) asm volatile("v_alignbit_b32 %0, %0, %1, 22" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 13) {
#define S(a) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 14) {
#define S(a) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 15) {
#define S(a) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 16) {
#define S(a) asm volatile("v_bfe_u32 %0, %0, 3, 15" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 17) {
#define S(a) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 18) {
#define S(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 19) {
#define S(a) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 20) {
#define S(a) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 21) {
#define S(a) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 22) {
#define S(a) asm volatile("v_lshrrev_b32_e64 %0, %0, %1" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 23) {
#define S(a) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 24) {
#define S(a) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 25) {
#define S(a) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 26) {
#define S(a) asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        // ---- multiplies ----
        } else if constexpr (OP == 27) {
#define S(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 28) {
#define S(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 29) {
#define S(a) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 30) {
#define S(q) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q) : "v"(a0), "s"(s) : "vcc");
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 31) {
#define S(q) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(a0), "s"(s) : "vcc");
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 32) {
#define S(q) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(a0), "v"(c) : "vcc");
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 33) {
#define S(a) asm volatile("v_mad_u32_u24 %0, %0, 37, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 34) {
#define S(a) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        // ---- 64-bit forms ----
        } else if constexpr (OP == 35) {
#define S(q) asm volatile("v_lshl_add_u64 %0, %1, 1, %0" : "+v"(q) : "v"(cq));
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 36) {
#define S(q) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q) : "s"(s));
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 37) {
#define S(q) asm volatile("v_lshrrev_b64 %0, 22, %0" : "+v"(q));
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 38) {
#define S(q) asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q), "v"(cq) : "vcc");
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 39) {
#define S(a) asm volatile("v_cmp_le_u32 vcc, %0, %1" :: "v"(a), "v"(c) : "vcc");
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 40) {
#define S(a) asm volatile("v_addc_co_u32_e64 %0, vcc, 0, 0, vcc" : "=v"(a) :: "vcc");
            R8(S) R8(S)
#undef S
        // ---- SDWA / DPP ----
        } else if constexpr (OP == 41) {
#define S(a) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 42) {
#define S(a) asm volatile("v_cmp_le_u32_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:WORD_0" :: "v"(a), "v"(c) : "vcc");
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 43) {
#define S(a) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 44) {
#define S(a) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(a) : "s20");
            R8(S) R8(S)
#undef S
        // ---- mixes: is the model additive?  what do scalar instructions and s_nop cost beside VALU? ----
        } else if constexpr (OP == 45) {                 // 8 cheap + 8 multiplies
#define S(a) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S)
#undef S
#define S(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S)
#undef S
        } else if constexpr (OP == 46) {                 // 16 cheap VALU + 16 SALU interleaved
#define S(a) asm volatile("v_xor_b32 %0, %1, %0\n\ts_add_u32 s20, s20, 1" : "+v"(a) : "v"(c) : "s20", "scc");
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 47) {                 // 16 VOP3 + 16 SALU interleaved
#define S(a) asm volatile("v_alignbit_b32 %0, %0, %1, 22\n\ts_add_u32 s20, s20, 1" : "+v"(a) : "v"(c) : "s20", "scc");
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 48) {                 // 16 cheap VALU + 16 s_nop 0
#define S(a) asm volatile("v_xor_b32 %0, %1, %0\n\ts_nop 0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 49) {                 // 16 VOP3 + 16 s_nop 0
#define S(a) asm volatile("v_alignbit_b32 %0, %0, %1, 22\n\ts_nop 0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 50) {                 // exec save / restore around a VALU op (process_word_defer's append)
#define S(a) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n\tv_xor_b32 %0, %1, %0\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(c) : "s20", "s21", "scc");
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0 | 0x80000000u), "v"(c | 0xC0000000u) : "vcc");
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 51) {                 // v_sad_u8 accumulate
#define S(a) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 52) {                 // ds_read_b32 random addresses: issue cost beside nothing else
#define S(a) asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"((c * 4u) & 0xFFFCu) : "memory");
            R8(S) R8(S)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef S
        } else if constexpr (OP == 53) {                 // 16 VOP3 + 4 ds_read (the deferring filter's ratio is ~28 : 1)
#define S(a) asm volatile("v_alignbit_b32 %0, %0, %1, 22" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
            uint32_t t0_, t1_, t2_, t3_;
            asm volatile("ds_read_b32 %0, %1" : "=v"(t0_) : "v"((a0 * 4u) & 0xFFFCu) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(t1_) : "v"((a1 * 4u) & 0xFFFCu) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(t2_) : "v"((a2 * 4u) & 0xFFFCu) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(t3_) : "v"((a3 * 4u) & 0xFFFCu) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            a4 ^= t0_ ^ t1_ ^ t2_ ^ t3_;
        } else if constexpr (OP == 54) {
#define S(a) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\tv_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(c) : "vcc");
            R8(S)
#undef S
        } else if constexpr (OP == 55) {
#define S(a) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 56) {
#define S(a) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 57) {                 // v_min3_u32 (VOP3)
#define S(a) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 58) {                 // v_xad_u32: (a ^ b) + c
#define S(a) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 59) {                 // dependent chain on ONE accumulator (latency, not issue)
#define S(a) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a0) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 60) {
#define S(a) asm volatile("v_alignbit_b32 %0, %0, %1, 22" : "+v"(a0) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 61) {
#define S(q) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(a0), "s"(s) : "vcc");
            Q8(S) Q8(S)
#undef S
        } else if constexpr (OP == 62) {                 // v_pk_mul_lo_u16 (VOP3P)
#define S(a) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 63) {                 // 64-bit xor as two VOP2 (hipcc's form)
#define S(a) asm volatile("v_xor_b32 %0, %1, %0\n\tv_xor_b32 %0, %2, %0" : "+v"(a) : "v"(c), "v"(d));
            R8(S)
#undef S
        } else if constexpr (OP == 64) {                 // strictly alternating cheap / slow (x m x m ...)
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a0), "+v"(a4) : "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a1), "+v"(a5) : "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a2), "+v"(a6) : "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a3), "+v"(a7) : "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a0), "+v"(a4) : "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a1), "+v"(a5) : "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a2), "+v"(a6) : "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a3), "+v"(a7) : "v"(c));
        } else if constexpr (OP == 65) {                 // pairs: x x m m x x m m ...
#define P(x0, x1, m0, m1) asm volatile("v_xor_b32 %0, %4, %0\n\tv_xor_b32 %1, %4, %1\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(m0), "+v"(m1) : "v"(c));
            P(a0, a1, a4, a5) P(a2, a3, a6, a7) P(a0, a1, a4, a5) P(a2, a3, a6, a7)
#undef P
        } else if constexpr (OP == 66) {                 // alternating cheap (scalar source) / slow
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a0), "+v"(a4) : "s"(s), "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a1), "+v"(a5) : "s"(s), "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a2), "+v"(a6) : "s"(s), "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a3), "+v"(a7) : "s"(s), "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a0), "+v"(a4) : "s"(s), "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a1), "+v"(a5) : "s"(s), "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a2), "+v"(a6) : "s"(s), "v"(c));
            asm volatile("v_xor_b32 %0, %2, %0\n\tv_mul_lo_u32 %1, %1, %3" : "+v"(a3), "+v"(a7) : "s"(s), "v"(c));
        } else if constexpr (OP == 67) {                 // the mad chain's shape: mad, mov, mad, mov (mov feeds nothing here)
#define S(q) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mov_b32 %1, %2" : "+v"(q), "=v"(a7) : "v"(a0), "s"(s) : "vcc");
            Q8(S)
#undef S
        } else if constexpr (OP == 68) {                 // 3 slow : 1 cheap
#define P(m0, m1, m2, x0) asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_xor_b32 %3, %4, %3" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(x0) : "v"(c));
            P(a0, a1, a2, a3) P(a4, a5, a6, a7) P(a0, a1, a2, a3) P(a4, a5, a6, a7)
#undef P
        } else if constexpr (OP == 69) {                 // 3 slow : 1 cheap with a scalar source
#define P(m0, m1, m2, x0) asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_xor_b32 %3, %5, %3" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(x0) : "v"(c), "s"(s));
            P(a0, a1, a2, a3) P(a4, a5, a6, a7) P(a0, a1, a2, a3) P(a4, a5, a6, a7)
#undef P
        } else if constexpr (OP == 70) {                 // v_or_b32, v_sub_u32, v_and v,v, v_max: which opcodes are in the cheap class?
#define S(a) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 71) {
#define S(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 72) {
#define S(a) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 73) {
#define S(a) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 74) {
#define S(a) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 75) {                 // v_cndmask with the mask in vcc, set once outside the loop
#define S(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c) : );
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 76) {                 // v_cndmask, mask in a scalar pair (VOP3)
#define S(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(c), "s"(sq));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 77) {                 // v_not / v_bfi / v_xnor
#define S(a) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(c), "v"(d));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 78) {                 // bitop3 with a scalar source
#define S(a) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(c), "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 79) {                 // bitop3 alternating with slow
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a0), "+v"(a4) : "v"(c), "v"(d));
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a1), "+v"(a5) : "v"(c), "v"(d));
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a2), "+v"(a6) : "v"(c), "v"(d));
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a3), "+v"(a7) : "v"(c), "v"(d));
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a0), "+v"(a4) : "v"(c), "v"(d));
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a1), "+v"(a5) : "v"(c), "v"(d));
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a2), "+v"(a6) : "v"(c), "v"(d));
            asm volatile("v_bitop3_b32 %0, %0, %2, %3 bitop3:0x96\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a3), "+v"(a7) : "v"(c), "v"(d));
        } else if constexpr (OP == 80) {                 // slow ops only, alternating two kinds (is 'slow' itself additive?)
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a0), "+v"(a4) : "v"(c));
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a1), "+v"(a5) : "v"(c));
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a2), "+v"(a6) : "v"(c));
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a3), "+v"(a7) : "v"(c));
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a0), "+v"(a4) : "v"(c));
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a1), "+v"(a5) : "v"(c));
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a2), "+v"(a6) : "v"(c));
            asm volatile("v_alignbit_b32 %0, %0, %2, 22\n\tv_mul_lo_u32 %1, %1, %2" : "+v"(a3), "+v"(a7) : "v"(c));
        } else if constexpr (OP == 81) {                 // the k > 16 canonical choice: v_cmp_lt_u64 -> vcc, two v_cndmask_b32 (VOP2, vcc)
#define S(q) asm volatile("v_cmp_lt_u64 vcc, %0, %3\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %5, vcc" : "+v"(q), "+v"(a0), "+v"(a1) : "v"(cq), "v"(c), "v"(d) : "vcc");
            Q8(S)
#undef S
        } else if constexpr (OP == 82) {                 // ... with the mask in a scalar pair and VOP3 selects
#define S(q) asm volatile("v_cmp_lt_u64 s[20:21], %0, %3\n\tv_cndmask_b32_e64 %1, %1, %4, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %5, s[20:21]" : "+v"(q), "+v"(a0), "+v"(a1) : "v"(cq), "v"(c), "v"(d) : "s20", "s21");
            Q8(S)
#undef S
        } else if constexpr (OP == 83) {                 // VOP2 reading vcc as carry-in
#define S(a) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(c) : "vcc");
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 84) {                 // v_cndmask VOP2 with distinct destination / sources
#define S(a) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(c), "v"(d) : );
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 85) {                 // cmp + cndmask_e64(0, 4) + add: the lane-private append's pointer update
#define S(a) asm volatile("v_cmp_le_u32_sdwa vcc, %0, %2 src0_sel:WORD_0 src1_sel:WORD_1\n\tv_cndmask_b32_e64 %1, 0, 4, vcc\n\tv_add_u32 %0, %0, %1" : "+v"(a), "=v"(q0) : "v"(c) : "vcc");
#undef S
            uint32_t t_;
#define S(a) asm volatile("v_cmp_le_u32_sdwa vcc, %0, %2 src0_sel:WORD_0 src1_sel:WORD_1\n\tv_cndmask_b32_e64 %1, 0, 4, vcc\n\tv_add_u32 %0, %0, %1" : "+v"(a), "=v"(t_) : "v"(c) : "vcc");
            R8(S)
#undef S
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    const uint64_t rq = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
    r ^= (uint32_t)rq ^ (uint32_t)(rq >> 32);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - rt0; }
}

static int g_waves = 4;

template <int OP>
static void run(const char *name, const char *cls, uint32_t *sink, int per_iter = 16)
{
    const int iters = 400000, blocks = 256 * g_waves;   // g_waves workgroups of 4 waves per CU = g_waves waves per SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    unsigned long long *clk, h[2];
    CHK(hipMalloc(&clk, 16));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 65536 / g_waves >= 16384 ? 16384 : 65536 / g_waves, 0, sink, iters / 8, 0x12345u, 0x1234567890ull, clk);
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 16384, 0, sink, iters, 0x12345u, 0x1234567890ull, clk);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);     // memrealtime ticks at 100 MHz
    const double ns = ms * 1e6 / ((double)g_waves * iters * per_iter);
    printf("%-52s %-24s %.3f ns per wave-instr per SIMD, clock %.2f GHz -> %5.2f cycles\n", name, cls, ns, ghz, ns * ghz);
    g_rows.push_back({name, cls, ns, ghz, per_iter});
    CHK(hipFree(clk));
}

int main(int argc, char **argv)
{
    const char *json = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--json") && i + 1 < argc) json = argv[++i];
        else if (!strcmp(argv[i], "--waves") && i + 1 < argc) g_waves = atoi(argv[++i]);
    }
    uint32_t *sink;
    CHK(hipMalloc(&sink, 256 * 8 * 256 * 4));
    printf("# %d waves per SIMD\n", g_waves);
    run<0>("v_xor_b32 v,v", "vop2 v,v", sink);
    run<6>("v_add_u32 v,v", "vop2 v,v:add", sink);
    run<8>("v_min_u32 v,v", "vop2 v,v:min", sink);
    run<5>("v_lshrrev_b32 v,v", "vop2 v,v:shift", sink);
    run<9>("v_mov_b32 v", "vop2 v,v:mov", sink);
    run<11>("v_cndmask_b32 v,v,vcc", "vop2 v,v:cndmask", sink);
    run<1>("v_xor_b32 s,v", "vop2 s,v", sink);
    run<56>("v_and_b32 s,v", "vop2 s,v:and", sink);
    run<2>("v_xor_b32 inline,v", "vop2 inline,v", sink);
    run<7>("v_add_u32 inline,v", "vop2 inline,v:add", sink);
    run<55>("v_lshlrev_b32 inline,v", "vop2 inline,v:shift", sink);
    run<4>("pair: v_lshrrev_b32 inline + v_add_u32 v,v (per instr)", "vop2 mix", sink);
    run<3>("v_xor_b32 literal,v", "vop2 literal,v", sink);
    run<10>("v_and_b32 literal,v", "vop2 literal,v:and", sink);
    run<63>("pair: v_xor v,v + v_xor v,v on one register (per instr)", "vop2 v,v:dep2", sink);
    run<12>("v_alignbit_b32 v,v,imm", "vop3", sink);
    run<13>("v_alignbit_b32 v,v,s", "vop3:alignbit s", sink);
    run<14>("v_perm_b32 v,v,s", "vop3:perm s", sink);
    run<15>("v_perm_b32 v,v,v", "vop3:perm v", sink);
    run<16>("v_bfe_u32 v,imm,imm", "vop3:bfe", sink);
    run<17>("v_bitop3_b32 v,v,v", "vop3:bitop3", sink);
    run<18>("v_and_or_b32 v,v,v", "vop3:and_or", sink);
    run<19>("v_lshl_or_b32 v,imm,v", "vop3:lshl_or", sink);
    run<20>("v_lshl_add_u32 v,imm,s", "vop3:lshl_add", sink);
    run<21>("v_add3_u32 v,v,v", "vop3:add3", sink);
    run<57>("v_min3_u32 v,v,v", "vop3:min3", sink);
    run<58>("v_xad_u32 v,v,v", "vop3:xad", sink);
    run<51>("v_sad_u8 v,v,v", "vop3:sad", sink);
    run<22>("v_lshrrev_b32_e64 v,s", "vop3 (simple op, e64)", sink);
    run<40>("v_addc_co_u32_e64 0,0,vcc", "vop3 (simple op, e64):addc", sink);
    run<23>("v_ffbh_u32 v", "vop3:ffbh", sink);
    run<24>("v_bcnt_u32_b32 v,v", "vop3:bcnt", sink);
    run<25>("v_mbcnt_lo_u32_b32 s,v", "v_mbcnt", sink);
    run<26>("v_mbcnt_hi_u32_b32 s,v", "v_mbcnt:hi", sink);
    run<27>("v_mul_lo_u32 v,s", "v_mul32", sink);
    run<28>("v_mul_lo_u32 v,v", "v_mul32:v,v", sink);
    run<29>("v_mul_hi_u32 v,s", "v_mul32:hi", sink);
    run<30>("v_mad_u64_u32 v,s,0", "v_mad_u64_u32:0", sink);
    run<31>("v_mad_u64_u32 v,s,pair", "v_mad_u64_u32", sink);
    run<32>("v_mad_u64_u32 v,v,pair", "v_mad_u64_u32:v,v", sink);
    run<33>("v_mad_u32_u24 v,inline,v", "v_mul24", sink);
    run<34>("v_mul_u32_u24 v,v (VOP2)", "v_mul24:vop2", sink);
    run<62>("v_pk_mul_lo_u16 v,v", "vop3p", sink);
    run<35>("v_lshl_add_u64 pair,1,pair", "v_lshl_add_u64", sink);
    run<36>("v_lshrrev_b64 s,pair", "v_shift64", sink);
    run<37>("v_lshrrev_b64 imm,pair", "v_shift64:imm", sink);
    run<38>("v_cmp_lt_u64 pair,pair", "v_cmp:u64", sink);
    run<39>("v_cmp_le_u32 v,v", "v_cmp", sink);
    run<54>("pair: v_sub_co_u32 + v_subb_co_u32 (per instr)", "vop2 carry", sink);
    run<41>("v_and_b32_sdwa WORD_1", "vop sdwa", sink);
    run<42>("v_cmp_le_u32_sdwa WORD_0,WORD_0", "vop sdwa:cmp", sink);
    run<43>("v_mov_b32_dpp row_shr:1", "vop dpp", sink);
    run<44>("v_readfirstlane_b32", "v_readlane", sink);
    run<45>("mix 8 v_xor v,v + 8 v_mul_lo v,v (per instr)", "mix:cheap+mul", sink);
    run<46>("v_xor v,v + s_add_u32 interleaved (per VALU instr)", "mix:cheap+salu", sink);
    run<47>("v_alignbit + s_add_u32 interleaved (per VALU instr)", "mix:vop3+salu", sink);
    run<48>("v_xor v,v + s_nop 0 interleaved (per VALU instr)", "mix:cheap+nop", sink);
    run<49>("v_alignbit + s_nop 0 interleaved (per VALU instr)", "mix:vop3+nop", sink);
    run<50>("saveexec; v_xor v,v; s_or exec (per VALU instr)", "mix:cheap+exec", sink);
    run<52>("ds_read_b32 random (per LDS instr, nothing else)", "ds_read", sink);
    run<53>("16 v_alignbit + 4 ds_read_b32 + wait (per VALU instr)", "mix:vop3+ds_read", sink);
    run<59>("v_xor_b32 v,v DEPENDENT chain", "latency:vop2", sink);
    run<60>("v_alignbit_b32 DEPENDENT chain", "latency:vop3", sink);
    run<61>("v_mad_u64_u32 DEPENDENT chain", "latency:mad64", sink);
    run<64>("alternating v_xor v,v / v_mul_lo (per instr)", "mix:alt cheap/slow", sink);
    run<65>("pairs x x m m (per instr)", "mix:pairs cheap/slow", sink);
    run<45>("blocked 8 x + 8 m (per instr)", "mix:blocked cheap/slow", sink);
    run<66>("alternating v_xor s,v / v_mul_lo (per instr)", "mix:alt cheap(s)/slow", sink);
    run<68>("3 v_mul_lo : 1 v_xor v,v (per instr)", "mix:3slow+1cheap", sink);
    run<69>("3 v_mul_lo : 1 v_xor s,v (per instr)", "mix:3slow+1cheap(s)", sink);
    run<67>("v_mad_u64_u32 + v_mov alternating (per instr)", "mix:mad+mov", sink);
    run<79>("alternating v_bitop3 / v_mul_lo (per instr)", "mix:alt bitop3/slow", sink);
    run<80>("alternating v_alignbit / v_mul_lo (per instr)", "mix:alt slow/slow", sink);
    run<70>("v_or_b32 v,v", "vop2 v,v:or", sink);
    run<71>("v_sub_u32 v,v", "vop2 v,v:sub", sink);
    run<72>("v_and_b32 v,v", "vop2 v,v:and", sink);
    run<73>("v_max_u32 v,v", "vop2 v,v:max", sink);
    run<74>("v_lshlrev_b32 v,v", "vop2 v,v:shl", sink);
    run<75>("v_cndmask_b32 v,v,vcc (vcc loop-invariant)", "vop2 v,v:cndmask2", sink);
    run<76>("v_cndmask_b32_e64 v,v,s[pair]", "vop3:cndmask", sink);
    run<77>("v_bfi_b32 v,v,v", "vop3:bfi", sink);
    run<78>("v_bitop3_b32 v,v,s", "vop3:bitop3 s", sink);
    run<81>("v_cmp_lt_u64 vcc + 2 v_cndmask_b32 VOP2 (per instr)", "mix:cmp64+cndmask", sink, 24);
    run<82>("v_cmp_lt_u64 s[] + 2 v_cndmask_b32_e64 (per instr)", "mix:cmp64+cndmask_e64", sink, 24);
    run<83>("v_addc_co_u32 VOP2 (vcc in/out)", "vop2 carry:addc", sink);
    run<84>("v_cndmask_b32 VOP2 dst != src", "vop2 v,v:cndmask3", sink);
    run<85>("cmp_sdwa + cndmask_e64(0,4) + add (per instr)", "mix:append ptr", sink, 24);
    if (json) {
        FILE *f = fopen(json, "w");
        if (!f) { perror(json); return 1; }
        fprintf(f, "{\n \"waves_per_simd\": %d,\n \"rows\": [\n", g_waves);
        for (size_t i = 0; i < g_rows.size(); ++i)
            fprintf(f, "  {\"name\": \"%s\", \"class\": \"%s\", \"ns\": %.4f, \"ghz\": %.3f, \"cycles\": %.3f}%s\n", g_rows[i].name, g_rows[i].cls,
                    g_rows[i].ns, g_rows[i].ghz, g_rows[i].ns * g_rows[i].ghz, i + 1 < g_rows.size() ? "," : "");
        fprintf(f, " ]\n}\n");
        fclose(f);
    }
    return 0;
}
