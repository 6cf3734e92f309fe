// tools/tsan_multictx.cpp — ThreadSanitizer stress of the C ABI's host side: several host threads, one lash_ctx each (the
// contract of include/lash_gfx950.h), calling the same entry points at the same time.  The library holds no process-global
// mutable state, so TSan must stay silent about lash_* frames.  Built and run by tools/tsan_multictx.sh (host code of the
// library instrumented with -fsanitize=thread).  Without a GPU the contexts cannot be created (LASH_ENODEV) and only the
// host-only entries are exercised; on a GPU box every thread also sketches small batches with LASH_TRACE_HOST set.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/lash_gfx950.h"

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 6, rounds = argc > 2 ? atoi(argv[2]) : 40;
    const int n_dev = lash_device_count();
    std::atomic<int> failures{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) {
        pool.emplace_back([&, t] {
            lash_ctx *ctx = nullptr;
            const int rc = lash_ctx_create(&ctx, n_dev > 0 ? t % n_dev : 0);
            if (n_dev > 0 && rc != LASH_OK) { failures++; return; }
            if (n_dev == 0 && rc != LASH_ENODEV) { failures++; return; }
            std::vector<uint8_t> seq(200000);
            for (size_t i = 0; i < seq.size(); ++i) seq[i] = "ACGT"[(i * 2654435761u + t * 40503u + (i >> 7)) & 3];
            seq[1000 + t] = 'N';
            const uint64_t rec_off[3] = {0, 120000, seq.size()}, goff[3] = {0, 1, 2};
            for (int r = 0; r < rounds; ++r) {
                lash_layout lay;
                if (lash_layout_parse(r & 1 ? "codes=ACTG,hmh_x=low" : "", &lay) != LASH_OK) failures++;
                lash_params prm{(r + t) % 3, 16 + (r % 5), 12, 0, 42};
                if (lash_params_check(&prm) != LASH_OK) failures++;
                const size_t ib = lash_layout_image_bytes(&lay, prm.algo, prm.p);
                std::vector<uint8_t> regs(4096, (uint8_t)(60 + (r & 7)));
                if (!(lash_ull_estimate(regs.data(), 12, r & 1) > 0.0)) failures++;
                (void)lash_strerror(-(r % 7));
                if (!ctx) continue;
                if (lash_ctx_set_layout(ctx, &lay) != LASH_OK) failures++;
                std::vector<uint8_t> img(2 * ib), img2(2 * ib);
                if (lash_sketch_batch(ctx, &prm, seq.data(), rec_off, 2, goff, 2, img.data()) != LASH_OK) failures++;
                if (lash_sketch_batch_async(ctx, &prm, seq.data(), rec_off, 2, goff, 2, img2.data()) != LASH_OK) failures++;
                if (lash_ctx_synchronize(ctx) != LASH_OK) failures++;
                if (memcmp(img.data(), img2.data(), img.size()) != 0) failures++;
            }
            if (ctx) lash_ctx_destroy(ctx);
        });
    }
    for (auto &th : pool) th.join();
    printf("tsan_multictx: %d threads x %d rounds, %d device(s), %d failures\n", T, rounds, n_dev, failures.load());
    return failures.load() ? 1 : 0;
}
