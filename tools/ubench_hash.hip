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
__global__ void __launch_bounds__(1024) hash_bench(unsigned long long *cycles, uint32_t *sink, int iters, uint64_t bitflip)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    uint32_t c0 = threadIdx.x * 2654435761u + blockIdx.x, c1 = c0 ^ 0x9E3779B9u;
    uint32_t acc = 0;
    if constexpr (MODE == 4) {      // desynchronise the waves: wave w of block b sleeps (w*7 + b*3) % 64 * 64 cycles first
        const int d = ((threadIdx.x >> 6) * 7 + blockIdx.x * 3) & 63;
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(1);
    }
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
            } else {
                uint64_t lo, hi;
                xxh3_128_4b(can, bitflip, lo, hi);
                if constexpr (MODE == 1) {        // + hash
                    acc ^= (uint32_t)lo ^ (uint32_t)(hi >> 32) ^ (uint32_t)hi;
                } else {                          // + register rule (+ atomic for MODE 3)
                    const uint32_t xh = (uint32_t)(hi >> 32), xl = (uint32_t)hi;
                    const uint32_t th = alignbit(xh, xl, 18);
                    const uint32_t reg = ((ffbh_u32(th) << 10) | ((uint32_t)lo & 0x3FFu)) + 0x400u;
                    const uint32_t bucket = xh >> 18;
                    if constexpr (MODE == 2) acc ^= reg + bucket;
                    else (void)__hip_atomic_fetch_max(lds + bucket, reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        c0 = c1; c1 = c1 * 1664525u + 1013904223u + acc;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x12345) sink[0] = acc + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
        cycles[65536 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = rt1 - rt0;
    }
}

template <int MODE>
void run(const char *name, int threads, int blocks_per_cu, int iters, unsigned long long *d_cyc, uint32_t *d_sink)
{
    const int blocks = 256 * blocks_per_cu;
    auto kern = hash_bench<MODE>;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 65536, 0, d_cyc, d_sink, iters / 4, 0xeef023344dc994d6ull);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 65536, 0, d_cyc, d_sink, iters, 0xeef023344dc994d6ull);
    CHK(hipDeviceSynchronize());
    int nw = blocks * threads / 64;
    std::vector<unsigned long long> h(nw), hr(nw);
    CHK(hipMemcpy(h.data(), d_cyc, nw * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hr.data(), d_cyc + 65536, nw * 8, hipMemcpyDeviceToHost));
    double avg = 0, avgr = 0; for (auto v : h) avg += (double)v; for (auto v : hr) avgr += (double)v;
    avg /= nw; avgr /= nw;
    const double waves_per_simd = (double)threads * blocks_per_cu / 256.0;
    const double cyc_per_kmer_wave = avg / ((double)iters * 16.0);
    printf("%-28s threads=%4d x%d/CU (%.0f waves/SIMD): %7.1f cycles per wave-kmer per wave, %6.1f per SIMD, clock %.2f GHz -> %.1f ns/wave-kmer/SIMD\n",
           name, threads, blocks_per_cu, waves_per_simd, cyc_per_kmer_wave, cyc_per_kmer_wave / waves_per_simd,
           avg / (avgr * 10.0), cyc_per_kmer_wave / waves_per_simd / (avg / (avgr * 10.0)));
}

int main()
{
    unsigned long long *d_cyc; uint32_t *d_sink;
    CHK(hipMalloc(&d_cyc, 2 * 65536 * 8)); CHK(hipMalloc(&d_sink, 4096));
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int threads = cfg == 0 ? 256 : 512, per_cu = cfg == 2 ? 2 : 1;
        run<0>("kmer extract", threads, per_cu, 2000, d_cyc, d_sink);
        run<1>("+ xxh3_128", threads, per_cu, 2000, d_cyc, d_sink);
        run<2>("+ register rule", threads, per_cu, 2000, d_cyc, d_sink);
        run<3>("+ ds_max_u32", threads, per_cu, 2000, d_cyc, d_sink);
    }
    run<4>("+ ds_max_u32, desynced", 512, 2, 2000, d_cyc, d_sink);
    run<3>("+ ds_max_u32, 10 rounds", 512, 20, 200, d_cyc, d_sink);
    return 0;
}
