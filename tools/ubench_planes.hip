// ubench_planes.hip — what bounds the bit-plane pair kernel (lash_amd/csrc/pair_planes.hip): the same kernel with its column
// (scalar) loads as they are / all hitting one scalar-cache line / removed, on random planes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lash_amd/csrc [-DLASH_PLANES_QT_FULL=.. -DLASH_PLANES_QT_GEN=..] -o tools/ubench_planes0 tools/ubench_planes.hip
#include "../lash_amd/csrc/pair_planes.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(uint32_t *p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        p[i] = x;
    }
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 16384, rows = argc > 2 ? atoi(argv[2]) : 2048;
    const uint32_t ld = (n + 511) / 512 * 512, npad = (n + 63) / 64 * 64;
    uint32_t *T, *S, *c, *m;
    CHK(hipMalloc(&T, lash::hmh_planes_T_words(ld) * 4));
    CHK(hipMalloc(&S, lash::hmh_planes_S_words(npad) * 4));
    CHK(hipMalloc(&c, (size_t)rows * n * 4));
    CHK(hipMalloc(&m, (size_t)rows * n * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, T, lash::hmh_planes_T_words(ld), 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, S, lash::hmh_planes_S_words(npad), 2u);
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int full = 1; full >= 0; --full) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0, 0));
            for (uint32_t r0 = 0; r0 < n; r0 += rows)
                CHK(lash::launch_hmh_pairs_planes(T, ld, r0, std::min(rows, n - r0), S, npad, n, full != 0, false, c, m, n, 0));
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("QT=%d/%d %s: %u x %u pairs in %.2f ms -> %.3g pairs/s\n", LASH_PLANES_QT_FULL, LASH_PLANES_QT_GEN, full ? "full sketches (17 instr)" : "general (20 instr)", n, n, ms,
                            (double)n * n / (ms * 1e-3));
        }
    }
    return 0;
}
