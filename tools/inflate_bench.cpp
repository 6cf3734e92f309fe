// tools/inflate_bench.cpp — MB/s of the host's gzip readers on one file: inflate_fast.hpp's decoder (+ CRC) against zlib.
// Build: g++ -O2 -std=c++17 -o tools/inflate_bench tools/inflate_bench.cpp lash_amd/csrc/host/inflate_fast.cpp -lz
// Run:   tools/inflate_bench FILE.gz [repeats]
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../lash_amd/csrc/host/inflate_fast.hpp"

using namespace lashhost;

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: inflate_bench FILE.gz [repeats]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    std::vector<uint8_t> src;
    uint8_t buf[1 << 16];
    size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) src.insert(src.end(), buf, buf + r);
    fclose(f);
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    double best_fast = 0, best_zlib = 0;
    size_t n_out = 0;
    for (int rep = 0; rep < reps; ++rep) {
        ByteSink s;
        auto t0 = std::chrono::steady_clock::now();
        const char *e = gunzip_members(src.data(), src.size(), s, false, nullptr);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (e) { fprintf(stderr, "fast decoder: %s\n", e); return 1; }
        n_out = s.n;
        best_fast = std::max(best_fast, s.n / dt / 1e6);
        // zlib: every member, into a buffer of the known size
        std::vector<uint8_t> out(n_out + 64);
        size_t in_at = 0, out_at = 0;
        t0 = std::chrono::steady_clock::now();
        while (in_at < src.size() && src[in_at] == 0x1f) {
            z_stream z = {};
            inflateInit2(&z, 31);
            z.next_in = src.data() + in_at; z.avail_in = (uInt)std::min<size_t>(src.size() - in_at, 1u << 30);
            z.next_out = out.data() + out_at; z.avail_out = (uInt)std::min<size_t>(out.size() - out_at, 1u << 30);
            const int rc = inflate(&z, Z_FINISH);
            in_at += z.total_in; out_at += z.total_out;
            inflateEnd(&z);
            if (rc != Z_STREAM_END) break;
        }
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (out_at != n_out) { fprintf(stderr, "zlib produced %zu bytes, fast decoder %zu\n", out_at, n_out); return 1; }
        best_zlib = std::max(best_zlib, out_at / dt / 1e6);
    }
    printf("%s: %zu -> %zu bytes; inflate_fast %.0f MB/s, zlib %.0f MB/s (x%.2f), one thread\n", argv[1], src.size(), n_out, best_fast, best_zlib,
           best_fast / best_zlib);
    return 0;
}
