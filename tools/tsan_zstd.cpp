// tools/tsan_zstd.cpp — ThreadSanitizer driver for the parallel-frames zstd writer (zstd_dl.cpp): streams of several sizes written in
// pieces of several sizes, read back, compared.  Built and run by tools/tsan_zstd.sh (CPU only).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../lash_amd/csrc/host/zstd_dl.hpp"

int main(int argc, char **argv)
{
    const std::string path = argc > 1 ? argv[1] : "/tmp/tsan_zstd.bin";
    const size_t sizes[] = {0, 1, (4u << 20), (16u << 20) + 7, (41u << 20) + 12345};
    const size_t pieces[] = {1u << 12, (1u << 20) + 3, 64u << 20};
    unsigned long long x = 88172645463325252ull;
    int bad = 0;
    for (size_t n : sizes)
        for (size_t piece : pieces) {
            std::vector<uint8_t> data(n);
            for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; data[i] = (uint8_t)((x >> 11) & 15u); }
            lashhost::ZstdWriter w;
            std::string e = w.open(path, 3, 4);
            for (size_t o = 0; e.empty() && o < n; o += piece) e = w.write(data.data() + o, std::min(piece, n - o));
            if (e.empty()) e = w.finish();
            std::vector<uint8_t> back;
            if (e.empty()) e = lashhost::zstd_decompress_file(path, back);
            if (!e.empty() || back != data) { fprintf(stderr, "MISMATCH n=%zu piece=%zu: %s\n", n, piece, e.c_str()); ++bad; }
        }
    printf("%s\n", bad ? "FAILED" : "ok: every stream read back as written");
    return bad ? 1 : 0;
}
