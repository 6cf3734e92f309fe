// tools/tsan_pgzip.cpp — ThreadSanitizer driver for the parallel gzip reader (host/pgzip.cpp + inflate_fast.cpp): reads a
// multi-member .gz with N inflate threads in odd-sized pieces and checks the bytes against a one-thread read.  CPU only.
// Build + run: tools/tsan_pgzip.sh
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../lash_amd/csrc/host/pgzip.hpp"

using namespace lashhost;

static bool slurp(const char *path, int threads, size_t piece, std::vector<uint8_t> &out)
{
    ParallelGzip pg;
    std::string err = pg.open(path, threads);
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return false; }
    std::vector<uint8_t> buf(piece);
    for (;;) {
        const long r = pg.read(buf.data(), buf.size(), err);
        if (r < 0) { fprintf(stderr, "%s\n", err.c_str()); return false; }
        if (r == 0) break;
        out.insert(out.end(), buf.begin(), buf.begin() + r);
    }
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::vector<uint8_t> want;
    if (!slurp(argv[1], 1, 1 << 20, want)) return 1;
    int bad = 0;
    for (int threads : {2, 8})
        for (size_t piece : {(size_t)4097, (size_t)1 << 22}) {
            std::vector<uint8_t> got;
            if (!slurp(argv[1], threads, piece, got) || got != want) { fprintf(stderr, "mismatch: %d threads, piece %zu\n", threads, piece); ++bad; }
        }
    printf("tsan_pgzip: %zu bytes, %d mismatches\n", want.size(), bad);
    return bad ? 1 : 0;
}
