// ubench.hip — instruction-rate microbenchmarks that price the sketch kernel's inner loop on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Prints steady-state ns (and cycles) per wave-instruction per SIMD at 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

enum Op { OP_XOR, OP_ALIGNBIT, OP_MUL_LO, OP_MUL_HI, OP_MAD_U64, OP_MUL_U24, OP_MAD_U24, OP_LSHR64, OP_ADD64, OP_FFBH, OP_BFE,
          OP_DS_MAX_RAND, OP_DS_MAX_SAME, OP_DS_OR_RAND, OP_DS_ADD_SEQ, OP_XOR3LIKE, OP_PERM, OP_MIX_XOR_MUL, OP_MIX_XOR_MAD64, OP_MIX_3XOR_MUL, OP_MOV, OP_ADD3, OP_LSHL_OR, OP_ADD_U32, OP_COUNT };
static const char *names[] = {"v_xor_b32", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24",
                              "v_mad_u32_u24", "v_lshrrev_b64", "v_lshl_add_u64", "v_ffbh_u32", "v_bfe_u32",
                              "ds_max_u32 random", "ds_max_u32 same-addr", "ds_or_b32 random", "ds_add_u32 lane-seq",
                              "v_bitop3/xor3", "v_perm_b32", "mix 8 xor + 8 mul_lo", "mix 8 xor + 8 mad_u64",
                              "mix 12 xor + 4 mul_lo", "v_mov_b32", "v_add3_u32", "v_lshl_or_b32", "v_add_u32"};

template <int OP>
__global__ void __launch_bounds__(1024) bench(unsigned long long *cycles, uint32_t *sink, int iters)
{
    extern __shared__ uint32_t lds[];            // 40 KiB: caps residency at 4 workgroups per CU
    if (OP >= OP_DS_MAX_RAND && OP <= OP_DS_ADD_SEQ) {
        for (int i = threadIdx.x; i < 10240; i += blockDim.x) lds[i] = 0;
        __syncthreads();
    }
    uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9E3779B9u, a2 = a0 * 3 + 7, a3 = a1 * 5 + 11;
    uint32_t a4 = a0 + 0x1234567, a5 = a1 + 0x7654321, a6 = a2 ^ 0xdeadbeef, a7 = a3 ^ 0xcafebabe;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
    const uint32_t c = 0x85EBCA97u + blockIdx.x;
    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define R8(STMT) STMT(a0, b0) STMT(a1, b1) STMT(a2, b2) STMT(a3, b3) STMT(a4, b4) STMT(a5, b5) STMT(a6, b6) STMT(a7, b7)
        if constexpr (OP == OP_XOR) {
#define S(a, b) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_ALIGNBIT) {
#define S(a, b) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_MUL_LO) {
#define S(a, b) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_MUL_HI) {
#define S(a, b) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_MAD_U64) {
#define S(a, b) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(b) : "v"(a), "v"(c) : "vcc");
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_MUL_U24) {
#define S(a, b) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_MAD_U24) {
#define S(a, b) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_LSHR64) {
#define S(a, b) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(b)); asm volatile("v_or_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S)
#undef S
#define S(a, b) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(b));
            R8(S)
#undef S
        } else if constexpr (OP == OP_ADD64) {
#define S(a, b) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(b));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_FFBH) {
#define S(a, b) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_BFE) {
#define S(a, b) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(a));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_XOR3LIKE) {
#define S(a, b) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(c), "v"(a7));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_MIX_XOR_MUL) {
#define S(a, b) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S)
#undef S
#define S(a, b) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b) : "v"(c));
            { uint32_t m0 = (uint32_t)b0, m1 = (uint32_t)b1, m2 = (uint32_t)b2, m3 = (uint32_t)b3, m4 = (uint32_t)b4, m5 = (uint32_t)b5, m6 = (uint32_t)b6, m7 = (uint32_t)b7;
              S(a0, m0) S(a1, m1) S(a2, m2) S(a3, m3) S(a4, m4) S(a5, m5) S(a6, m6) S(a7, m7)
              b0 = m0; b1 = m1; b2 = m2; b3 = m3; b4 = m4; b5 = m5; b6 = m6; b7 = m7; }
#undef S
        } else if constexpr (OP == OP_MIX_XOR_MAD64) {
#define S(a, b) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S)
#undef S
#define S(a, b) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(b) : "v"(a7), "v"(c) : "vcc");
            R8(S)
#undef S
        } else if constexpr (OP == OP_MIX_3XOR_MUL) {
#define S(a, b) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(c));
            R8(S) S(a0, b0) S(a1, b1) S(a2, b2) S(a3, b3)
#undef S
            { uint32_t m0 = (uint32_t)b0, m1 = (uint32_t)b1, m2 = (uint32_t)b2, m3 = (uint32_t)b3;
              asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(m0) : "v"(c)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(m1) : "v"(c));
              asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(m2) : "v"(c)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(m3) : "v"(c));
              b0 = m0; b1 = m1; b2 = m2; b3 = m3; }
        } else if constexpr (OP == OP_MOV) {
#define S(a, b) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_ADD3) {
#define S(a, b) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(a7));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_LSHL_OR) {
#define S(a, b) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_ADD_U32) {
#define S(a, b) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_PERM) {
#define S(a, b) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(a7));
            R8(S) R8(S)
#undef S
        } else if constexpr (OP == OP_DS_MAX_RAND || OP == OP_DS_OR_RAND) {
            // 16 LDS atomics per iteration on pseudo-random dword addresses (xorshift per lane), no return value
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 ^= a0 << 13; a0 ^= a0 >> 17; a0 ^= a0 << 5;
                uint32_t addr = (a0 >> 8) & (8191u << 2);
                if constexpr (OP == OP_DS_MAX_RAND) asm volatile("ds_max_u32 %0, %1" :: "v"(addr), "v"(a0) : "memory");
                else asm volatile("ds_or_b32 %0, %1" :: "v"(addr), "v"(a0) : "memory");
            }
        } else if constexpr (OP == OP_DS_MAX_SAME) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 ^= a0 << 13; a0 ^= a0 >> 17; a0 ^= a0 << 5;
                uint32_t addr = (uint32_t)j * 4;
                asm volatile("ds_max_u32 %0, %1" :: "v"(addr), "v"(a0) : "memory");
            }
        } else if constexpr (OP == OP_DS_ADD_SEQ) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 ^= a0 << 13; a0 ^= a0 >> 17; a0 ^= a0 << 5;
                uint32_t addr = ((threadIdx.x & 63) * 4 + j * 256) & 65535u;
                asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(a0) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) ^ (uint32_t)((b0 ^ b7) >> 32);
    if (r == 0x12345) sink[0] = r + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        const unsigned slot = (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) & 4095u;
        cycles[slot] = t1 - t0;
        cycles[4096 + slot] = rt1 - rt0;     // 100 MHz ticks
    }
}

// Steady-state pricing: 16 rounds of workgroups per CU slot (so start/finish skew averages out), 4 waves per SIMD
// (4 x 256-thread workgroups per CU, enforced with 40 KiB of dynamic LDS each), wall time from HIP events over a
// launch of several ms, clock from s_memtime / s_memrealtime inside the kernel.
template <int OP>
void run(int, int iters, unsigned long long *d_cyc, uint32_t *d_sink)
{
    const int threads = 256, blocks = 256 * 4 * 16;
    auto kern = bench<OP>;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 40960, 0, d_cyc, d_sink, iters / 4);   // warm-up
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 40960, 0, d_cyc, d_sink, iters);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = 4096;
    std::vector<unsigned long long> h(nw), hr(nw);
    CHK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hr.data(), d_cyc + 4096, nw * 8, hipMemcpyDeviceToHost));
    double avg = 0, avgr = 0; for (auto v : h) avg += (double)v; for (auto v : hr) avgr += (double)v;
    const double ghz = avg / (avgr * 10.0);                                   // shader cycles per ns
    const double instr_per_simd = (double)iters * 16.0 * ((double)blocks * threads / 64.0) / 1024.0;
    const double ns = (double)ms * 1e6 / instr_per_simd;
    printf("%-24s %7.3f ms  %6.3f ns per wave-instr per SIMD   clock %.2f GHz -> %5.2f cycles\n", names[OP], ms, ns, ghz, ns * ghz);
}

template <int OP> void both(unsigned long long *c, uint32_t *s) { run<OP>(0, OP >= OP_DS_MAX_RAND && OP <= OP_DS_ADD_SEQ ? 100 : 1600, c, s); }

__global__ void copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ void read_kernel(const uint4 *__restrict__ in, uint32_t *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567) out[0] = acc;
}

// the direct sketch kernel's access pattern: a lane owns 64 contiguous bytes (4 x dwordx4 at a 64-B lane stride) and
// also reads the 16 bytes after them; calibrates FETCH_SIZE for profiles/traffic.json
__global__ void read64_kernel(const uint4 *__restrict__ in, uint32_t *out, size_t n64)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + 1 < n64; i += stride) {
        const uint4 *p = in + 4 * i;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
        acc ^= a.x ^ b.y ^ c.z ^ d.w ^ e.x;
    }
    if (acc == 0x1234567) out[0] = acc;
}

int main(int argc, char **argv)
{
    const bool hbm_only = argc > 1 && !strcmp(argv[1], "hbm");
    unsigned long long *d_cyc; uint32_t *d_sink;
    CHK(hipMalloc(&d_cyc, 2 * 4096 * 8)); CHK(hipMalloc(&d_sink, 4096));
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
    if (!hbm_only) {
    both<OP_XOR>(d_cyc, d_sink); both<OP_ALIGNBIT>(d_cyc, d_sink); both<OP_MUL_LO>(d_cyc, d_sink); both<OP_MUL_HI>(d_cyc, d_sink);
    both<OP_MAD_U64>(d_cyc, d_sink); both<OP_MUL_U24>(d_cyc, d_sink); both<OP_MAD_U24>(d_cyc, d_sink); both<OP_LSHR64>(d_cyc, d_sink);
    both<OP_ADD64>(d_cyc, d_sink); both<OP_FFBH>(d_cyc, d_sink); both<OP_BFE>(d_cyc, d_sink); both<OP_XOR3LIKE>(d_cyc, d_sink);
    both<OP_PERM>(d_cyc, d_sink);
    both<OP_MOV>(d_cyc, d_sink); both<OP_ADD_U32>(d_cyc, d_sink); both<OP_ADD3>(d_cyc, d_sink); both<OP_LSHL_OR>(d_cyc, d_sink);
    both<OP_MIX_XOR_MUL>(d_cyc, d_sink); both<OP_MIX_XOR_MAD64>(d_cyc, d_sink); both<OP_MIX_3XOR_MUL>(d_cyc, d_sink);
    both<OP_DS_MAX_RAND>(d_cyc, d_sink); both<OP_DS_MAX_SAME>(d_cyc, d_sink); both<OP_DS_OR_RAND>(d_cyc, d_sink); both<OP_DS_ADD_SEQ>(d_cyc, d_sink);
    }
    // HBM copy / read bandwidth (2 GiB buffers, beyond the 256 MiB Infinity Cache)
    size_t bytes = (size_t)2 << 30;
    uint4 *a, *b; CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, a, b, bytes / 16);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("copy  2 GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(read_kernel, dim3(256 * 8), dim3(256), 0, 0, a, d_sink, bytes / 16);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("read  2 GiB: %.3f ms  %.2f TB/s\n", ms, (double)bytes / ms / 1e9);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(read64_kernel, dim3(256 * 8), dim3(256), 0, 0, a, d_sink, bytes / 64);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("read64 2 GiB (64 B + 16 B look-ahead per lane): %.3f ms  %.2f TB/s\n", ms, (double)bytes / ms / 1e9);
    }
    return 0;
}
