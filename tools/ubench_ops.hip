// ubench_ops.hip — issue cost of the bit-plane pair kernel's candidate instructions (operand classes matter), gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_ops tools/ubench_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) bench(uint32_t *sink, int iters, uint32_t sarg, unsigned long long *clk)
{
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9E3779B9u, a2 = a0 * 3 + 7, a3 = a1 * 5 + 11;
    uint32_t a4 = a0 + 0x1234567, a5 = a1 + 0x7654321, a6 = a2 ^ 0xdeadbeef, a7 = a3 ^ 0xcafebabe;
    const uint32_t c = 0x85EBCA97u + threadIdx.x;
    uint32_t s = sarg;                                   // wave-uniform -> SGPR
    for (int i = 0; i < iters; ++i) {
#define R8(STMT) STMT(a0) STMT(a1) STMT(a2) STMT(a3) STMT(a4) STMT(a5) STMT(a6) STMT(a7)
        if constexpr (OP == 0) {
#define S(a) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(c), "v"(a7));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 1) {
#define S(a) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xde" : "+v"(a) : "v"(c), "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 2) {
#define S(a) asm volatile("v_bitop3_b32 %0, %2, %0, %1 bitop3:0xde" : "+v"(a) : "v"(c), "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 3) {
#define S(a) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "s"(s));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 4) {
#define S(a) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 5) {
#define S(a) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(a7));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 6) {
#define S(a) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 7) {                  // xor with SGPR, then or into accumulator: 2 VOP2 per plane
#define S(a) asm volatile("v_xor_b32 %0, %2, %1\n\tv_or_b32 %3, %0, %3" : "=&v"(a6), "+v"(a7) : "s"(s), "v"(a) : );
            S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a0) S(a1)
#undef S
        } else if constexpr (OP == 8) {
#define S(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(a7));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 9) {
#define S(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(a7));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 10) {                 // bitop3, two distinct VGPRs only (src0 == dst, src1 == src2)
#define S(a) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 11) {                 // v_pk (VOP3P) or: packed 16-bit or as a plain 32-bit or
#define S(a) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == 12) {                 // 64-bit xor on register pairs
            uint64_t b0 = a0, b1 = a1;
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a0) : "v"(c));
        }
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - rt0; }
}

template <int OP>
static void run(const char *name, uint32_t *sink, int per_iter = 16)
{
    const int iters = 1000000, blocks = 256 * 4;        // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    unsigned long long *clk, h[2];
    CHK(hipMalloc(&clk, 16));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, sink, iters, 0x12345u, clk);
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, sink, iters, 0x12345u, clk);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);     // memrealtime ticks at 100 MHz
    // per SIMD: 4 waves x iters x per_iter instructions
    const double ns = ms * 1e6 / (4.0 * iters * per_iter);
    printf("%-44s %.3f ns per wave-instr per SIMD, clock %.2f GHz -> %.2f cycles\n", name, ns, ghz, ns * ghz);
}

int main()
{
    uint32_t *sink;
    CHK(hipMalloc(&sink, 256 * 4 * 256 * 4));
    run<6>("v_xor_b32 v,v", sink);
    run<3>("v_xor_b32 s,v", sink);
    run<0>("v_bitop3 v,v,v", sink);
    run<10>("v_bitop3 v,v,v (2 distinct sources)", sink);
    run<1>("v_bitop3 v,v,s (src2 sgpr)", sink);
    run<2>("v_bitop3 s,v,v (src0 sgpr)", sink);
    run<4>("v_bcnt_u32_b32 v,v", sink);
    run<5>("v_or3_b32 v,v,v", sink);
    run<8>("v_and_or_b32 v,v,v", sink);
    run<9>("v_fma_f32 v,v,v", sink);
    run<11>("v_pk_max_u16 v,v", sink);
    run<7>("pair: v_xor s,v + v_or v,v (per pair)", sink, 8);
    return 0;
}
