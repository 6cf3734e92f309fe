// ubench_hash.hip — in-register cost of the sketch kernel's per-k-mer work (no HBM traffic), to separate
// VALU issue time from everything else.  Variants: hash only; hash + register rule; + LDS atomic.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lash_amd/csrc -o tools/ubench_hash tools/ubench_hash.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lash_device.h"
using namespace lash;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) hash_bench(unsigned long long *cycles, uint32_t *sink, int iters, uint64_t bitflip64)
{
    const BitFlip bitflip = BitFlip::vector(bitflip64);
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    uint32_t c0 = threadIdx.x * 2654435761u + blockIdx.x, c1 = c0 ^ 0x9E3779B9u;
    uint32_t acc = 0;

    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        const uint32_t r0 = rcword(c0), r1 = rcword(c1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            uint32_t fwd = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
            uint32_t rc = r ? alignbit(r1, r0, 2 * r) : r0;
            uint32_t can = fwd < rc ? fwd : rc;
            if constexpr (MODE == 0) {            // k-mer extraction only
                acc ^= can;
            } else if constexpr (MODE == 1) {     // + the full xxh3_128 of 4 bytes (both halves, every bit)
                uint64_t lo, hi;
                xxh3_128_4b(can, bitflip, lo, hi);
                acc ^= (uint32_t)lo ^ (uint32_t)(hi >> 32) ^ (uint32_t)hi;
            } else if constexpr (MODE == 4) {     // hll p=14 k=21 (BASELINE configs[2]): 64-bit windows + xxh3_64 + rule + ds_max
                const uint32_t c2 = c1 * 0x9E3779B1u, r2 = rcword(c2);
                // the kernel's canon_gt16<21>: the window's halves as fields at compile-time places, the minimum through a scalar-pair mask
                constexpr int D = 10;
                const int s = 2 * r + D;
                const uint32_t f_lo = s < 32 ? alignbit(c0, c1, 32 - s) : s == 32 ? c1 : alignbit(c1, c2, 64 - s);
                const uint32_t f_hi = s <= 32 ? (uint32_t)__builtin_amdgcn_ubfe(c0, 32 - s, D) : alignbit(c0, c1, 32 - 2 * r) >> (32 - D);
                const uint32_t q_lo = r ? alignbit(r1, r0, 2 * r) : r0;
                const uint32_t q_hi = s <= 32 ? (uint32_t)__builtin_amdgcn_ubfe(r1, 2 * r, D) : alignbit(r2, r1, 2 * r) & ((1u << D) - 1u);
                uint32_t cn_lo, cn_hi;
                min_u64(f_lo, f_hi, q_lo, q_hi, cn_lo, cn_hi);
                const uint64_t cn = ((uint64_t)cn_hi << 32) | cn_lo;
                const uint64_t hh64 = xxh3_64_8b_pre((uint32_t)cn, (uint32_t)(cn >> 32), bitflip);   // (the kernel's fast form: add_kmer<1, FAST>)
                const uint32_t hh = (uint32_t)(hh64 >> 32), hl = (uint32_t)hh64;
                asm volatile("ds_max_i32 %0, %1" ::"v"(((hl ^ alignbit(hh, hl, 28)) & 16383u) << 2), "v"(ffbh_u32(hh)) : "memory");   // round 4: rho - 1, signed
            } else if constexpr (MODE == 5) {     // ull p=12 k=16 (configs[4]): xxh3_64 + rule + ds_or
                const uint64_t hh64 = xxh3_64_8b_pre(can, 0u, bitflip);                             // (add_kmer<2, FAST>)
                const uint32_t hh = (uint32_t)(hh64 >> 32), hl = (uint32_t)hh64;
                const uint32_t th = alignbit(hh, hl, 32 - 12) ^ (hh >> (28 - 12));
                asm volatile("ds_or_b32 %0, %1" ::"v"((hh >> 17) & ~7u), "v"((th < 1u ? th : 1u) << (ffbh_u32(th) & 31u)) : "memory");   // round 4: nlz bitmap, first word
            } else if constexpr (MODE == 6) {     // deferred signatures, step 1: the rank half of the hash alone
                acc ^= xxh3_128_4b_hmh_rank(can, bitflip);
            } else {                              // the sketch kernel's own fast path (add_kmer<HMH, x = high half, FAST>):
                uint32_t xh, sig;                 // only the bits the register rule reads, + the rule (+ the LDS atomic, MODE 3)
                xxh3_128_4b_hmh_fast(can, bitflip, xh, sig);
                const uint32_t t18 = (xh << 14) | 0x3FFFu;
                const uint32_t reg = (ffbh_u32(t18) << 10) | sig;                                  // round 4: (lz - 1) << 10 | sig, signed max
                const uint32_t bucket = xh >> 18;
                if constexpr (MODE == 2) acc ^= reg + bucket;
                else asm volatile("ds_max_i32 %0, %1" ::"v"(bucket << 2), "v"(reg) : "memory");
            }
        }
        c0 = c1; c1 = c1 * 1664525u + 1013904223u + acc;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x12345) sink[0] = acc + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        const unsigned slot = (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) & 8191u;
        cycles[slot] = t1 - t0;
        cycles[65536 + slot] = rt1 - rt0;
    }
}

// The deferring kernel's stream (sketch_kernels.hip, process_word_defer) in the same harness: groups of four k-mers — rank half,
// threshold word read back, test — STAGE 0; + push on the lane's stack — STAGE 1; + the drain rounds (full update, ds_min) — STAGE 2 (all
// of it).  The table fills as in a real work item (iters * 16 * 512 k-mers into 16 384 buckets per workgroup).
__device__ __forceinline__ uint32_t ub_lds_load(uint32_t b) { return *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)b; }
__device__ __forceinline__ void ub_lds_store(uint32_t b, uint32_t v) { *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)b = v; }
template <int STAGE>
__global__ void __launch_bounds__(1024) defer_bench(unsigned long long *cycles, uint32_t *sink, int iters, uint64_t bitflip64)
{
    // round 4 (sketch_kernels.hip: LdsThrRegs, SigQueue, process_word_defer): the table word is the threshold, every lane keeps a stack
    const BitFlip bitflip = BitFlip::vector(bitflip64);
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t c0 = threadIdx.x * 2654435761u + blockIdx.x, c1 = c0 ^ 0x9E3779B9u;
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t DEPTH = 7;
    const uint32_t lane_b = 65536u + (threadIdx.x >> 6) * 1792u + lane * DEPTH * 4u, lim = lane_b + 4u * (DEPTH - 4u);
    uint32_t ptr = lane_b;
    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        const uint32_t r0 = rcword(c0), r1 = rcword(c1);
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            uint32_t can[4], x16[4], cur[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = g + j;
                const uint32_t fwd = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
                const uint32_t rc = r ? alignbit(r1, r0, 2 * r) : r0;
                can[j] = fwd < rc ? fwd : rc;
                const uint32_t xh = xxh3_128_4b_hmh_rank(can[j], bitflip);
                cur[j] = ub_lds_load((xh >> 16) & 0xFFFCu);
                x16[j] = xh >> 2;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (STAGE == 0) { acc += (x16[j] & 0xFFFFu) <= (cur[j] >> 16); continue; }
                ub_lds_store(ptr, can[j]);
                uint32_t inc;
                asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:WORD_0 src1_sel:WORD_1\n\tv_cndmask_b32_e64 %0, 0, 4, vcc" : "=v"(inc) : "v"(x16[j]), "v"(cur[j]) : "vcc");
                ptr += inc;
            }
            if constexpr (STAGE >= 1) {
                if (__builtin_amdgcn_ballot_w64(ptr > lim) != 0ull) {
                    do {
                        if (ptr != lane_b) {
                            ptr -= 4u;
                            if constexpr (STAGE == 2) {
                                const uint32_t c = ub_lds_load(ptr);
                                uint32_t xh, sig;
                                xxh3_128_4b_hmh_fast(c, bitflip, xh, sig);
                                const uint32_t lzm1 = ffbh_u32((xh << 14) | 0x3FFFu);
                                asm volatile("ds_min_u32 %0, %1" ::"v"((xh >> 18) << 2), "v"(((0xFFFF0000u >> lzm1) & 0xFFFF0000u) | (0xFFFEu - sig)) : "memory");
                            }
                        }
                    } while (__builtin_amdgcn_ballot_w64(ptr > lim) != 0ull);
                }
            }
        }
        c0 = c1; c1 = c1 * 1664525u + 1013904223u + acc;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x12345) sink[0] = acc + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        const unsigned slot = (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) & 8191u;
        cycles[slot] = t1 - t0;
        cycles[65536 + slot] = rt1 - rt0;
    }
}

// Steady state: 20 rounds of 512-thread workgroups per CU (64 KiB of LDS each -> 2 resident per CU = 4 waves per
// SIMD, as in the sketch kernel), wall time from HIP events; the clock comes from s_memtime / s_memrealtime.
// (Per-wave in-kernel timing divided by the nominal waves per SIMD under-states the cost: waves are resident for
// only ~2/3 of a single-round launch.)
typedef void (*bench_kernel)(unsigned long long *, uint32_t *, int, uint64_t);
void run_kernel(bench_kernel kern, const char *name, int iters, unsigned long long *d_cyc, uint32_t *d_sink, unsigned lds_bytes = 65536);
template <int MODE>
void run(const char *name, int iters, unsigned long long *d_cyc, uint32_t *d_sink) { run_kernel(hash_bench<MODE>, name, iters, d_cyc, d_sink); }
void run_kernel(bench_kernel kern, const char *name, int iters, unsigned long long *d_cyc, uint32_t *d_sink, unsigned lds_bytes)
{
    const int threads = 512, blocks = 256 * 2 * 10;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds_bytes, 0, d_cyc, d_sink, iters / 4, 0xeef023344dc994d6ull);
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds_bytes, 0, d_cyc, d_sink, iters, 0xeef023344dc994d6ull);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = 8192;
    std::vector<unsigned long long> h(nw), hr(nw);
    CHK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hr.data(), d_cyc + 65536, nw * 8, hipMemcpyDeviceToHost));
    double avg = 0, avgr = 0; for (auto v : h) avg += (double)v; for (auto v : hr) avgr += (double)v;
    const double ghz = avg / (avgr * 10.0);
    const double wave_kmers_per_simd = (double)iters * 16.0 * ((double)blocks * threads / 64.0) / 1024.0;
    const double ns = (double)ms * 1e6 / wave_kmers_per_simd;
    printf("%-20s %7.3f ms  %6.1f ns per wave-k-mer per SIMD  clock %.2f GHz -> %6.1f cycles  (%.3g k-mers/s chip-wide)\n",
           name, ms, ns, ghz, ns * ghz, (double)iters * 16.0 * blocks * threads / (ms * 1e-3));
}

int main()
{
    unsigned long long *d_cyc; uint32_t *d_sink;
    CHK(hipMalloc(&d_cyc, 2 * 65536 * 8)); CHK(hipMalloc(&d_sink, 4096));
    run<0>("k-mer extract", 400, d_cyc, d_sink);
    run<1>("+ xxh3_128", 200, d_cyc, d_sink);
    run<2>("+ register rule", 200, d_cyc, d_sink);
    run<3>("+ ds_max_u32", 200, d_cyc, d_sink);
    run<4>("hll p14 k21 stream", 200, d_cyc, d_sink);
    run<5>("ull p12 k16 stream", 200, d_cyc, d_sink);
    // deferred signatures (HyperMinHash, long work items): 600 iterations = 300 k-mers per bucket, a 5 Mbp work item's load
    run<6>("hmh rank half only", 200, d_cyc, d_sink);
    run_kernel(defer_bench<0>, "defer: + read + test", 600, d_cyc, d_sink, 65536 + 14336);
    run_kernel(defer_bench<1>, "defer: + append", 600, d_cyc, d_sink, 65536 + 14336);
    run_kernel(defer_bench<2>, "defer: hmh k16 stream", 600, d_cyc, d_sink, 65536 + 14336);
    run<3>("+ ds_max_u32 (600)", 600, d_cyc, d_sink);
    return 0;
}
