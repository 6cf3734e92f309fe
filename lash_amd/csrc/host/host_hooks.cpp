// host_hooks.cpp — extern "C" test hooks over the host-side C++ (FASTX reader, JSON writers, zstd, list files) so
// that tests/ can drive them through ctypes.  Not part of the product ABI (that is include/lash_gfx950.h).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include <atomic>
#include <thread>

#include "dist_format.hpp"
#include "fastx.hpp"
#include "inflate_fast.hpp"
#include "json_out.hpp"
#include "name_order.hpp"
#include "pgzip.hpp"
#include "sketch_files.hpp"
#include "zstd_dl.hpp"

using namespace lashhost;

namespace {

// The test-only environment knobs of the host side are read HERE — in the library the tests load — and nowhere in the product objects
// (VERDICT r5 next #8): LASH_TEST_FAST_INFLATE_FAIL_AFTER=<bytes>, LASH_TEST_FAST_INFLATE_FLIP_AT=<output byte>
struct TestSeamsFromEnv {
    TestSeamsFromEnv()
    {
        if (const char *e = getenv("LASH_TEST_FAST_INFLATE_FAIL_AFTER")) lashhost::test_seams::inflate_fail_after = atol(e);
        if (const char *e = getenv("LASH_TEST_FAST_INFLATE_FLIP_AT")) lashhost::test_seams::inflate_flip_at = atol(e);
    }
} test_seams_from_env;

char *dup_str(const std::string &s)
{
    char *p = (char *)malloc(s.size() + 1);
    memcpy(p, s.c_str(), s.size() + 1);
    return p;
}
}  // namespace

extern "C" {

void lash_host_free(void *p) { free(p); }

// Parses a FASTA/FASTQ(.gz/.zst) file.  On success returns NULL and fills seq/rec_off (malloc'd, caller frees with
// lash_host_free); otherwise returns the error text (malloc'd).
char *lash_host_read_fastx(const char *path, uint8_t **seq, uint64_t *seq_bytes, uint64_t **rec_off, uint64_t *n_rec)
{
    RecordBatch rb;
    std::string err = read_fastx_file(path, rb);
    if (!err.empty()) return dup_str(err);
    *seq_bytes = rb.seq.size();
    *n_rec = rb.n_rec();
    *seq = (uint8_t *)malloc(rb.seq.size() + 1);
    if (!rb.seq.empty()) memcpy(*seq, rb.seq.data(), rb.seq.size());
    *rec_off = (uint64_t *)malloc(rb.rec_off.size() * 8);
    memcpy(*rec_off, rb.rec_off.data(), rb.rec_off.size() * 8);
    return nullptr;
}

// '\n'-separated items in, pretty JSON array out (malloc'd)
char *lash_host_json_array(const char *items_nl, uint64_t n_items)
{
    std::vector<std::string> v;
    const char *p = items_nl;
    for (uint64_t i = 0; i < n_items; ++i) {
        const char *e = strchr(p, '\n');
        v.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : p + strlen(p);
    }
    return dup_str(json_pretty_string_array(v));
}

// name_order.hpp: XXH3-64 with a seed, and the hashbrown key order of '\n'-separated names (order_out holds n_items
// slots; returns the number of distinct names written)
uint64_t lash_host_xxh3_64(const uint8_t *p, uint64_t n, uint64_t seed) { return xxh3_64_seeded(p, n, seed); }

uint64_t lash_host_name_order(const char *items_nl, uint64_t n_items, uint64_t seed, uint32_t *order_out)
{
    std::vector<std::string> v;
    const char *p = items_nl;
    for (uint64_t i = 0; i < n_items; ++i) {
        const char *e = strchr(p, '\n');
        v.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
        p = e ? e + 1 : p + strlen(p);
    }
    const std::vector<uint32_t> o = hashbrown_key_order(v, seed);
    std::copy(o.begin(), o.end(), order_out);
    return o.size();
}

// inflate_fast.hpp: every member of a gzip image; returns NULL and the malloc'd bytes, else the error text (malloc'd)
char *lash_host_gunzip(const uint8_t *src, uint64_t n, uint8_t **out, uint64_t *out_bytes)
{
    ByteSink sink;
    const char *e = gunzip_members(src, n, sink, false, nullptr);
    if (e) return dup_str(e);
    *out_bytes = sink.n;
    *out = (uint8_t *)malloc(sink.n + 1);
    if (sink.n) memcpy(*out, sink.p, sink.n);
    return nullptr;
}

uint32_t lash_host_crc32(uint32_t crc, const uint8_t *p, uint64_t n) { return crc32_fast(crc, p, n); }

// one gzip member through WindowedInflate (the sequential reader's shape: `window` bytes of room after 32 KiB of history),
// drained in pieces of `piece` bytes; checks the trailer like the reader does
char *lash_host_gunzip_windowed(const uint8_t *src, uint64_t n, uint64_t window, uint64_t piece, uint8_t **out, uint64_t *out_bytes,
                                uint64_t *consumed)
{
    const size_t hl = gzip_header_length(src, n);
    if (!hl) return dup_str("bad gzip header");
    WindowedInflate w(window);
    w.begin();
    std::vector<uint8_t> all, buf(piece ? piece : 1);
    size_t ip = hl;
    const char *err = nullptr;
    for (;;) {
        const long r = w.read(src, n, ip, buf.data(), buf.size(), &err);
        if (r < 0) return dup_str(err);
        if (r == 0) break;
        all.insert(all.end(), buf.begin(), buf.begin() + r);
    }
    if (ip + 8 > n) return dup_str("truncated gzip stream");
    const uint32_t want_crc = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
    const uint32_t want_len = (uint32_t)src[ip + 4] | ((uint32_t)src[ip + 5] << 8) | ((uint32_t)src[ip + 6] << 16) | ((uint32_t)src[ip + 7] << 24);
    if (w.crc() != want_crc || (uint32_t)w.total() != want_len || w.total() != all.size()) return dup_str("gzip trailer check failed");
    *consumed = ip + 8;
    *out_bytes = all.size();
    *out = (uint8_t *)malloc(all.size() + 1);
    if (!all.empty()) memcpy(*out, all.data(), all.size());
    return nullptr;
}


// pgzip.hpp: inflates a (multi-member) gzip file with `threads` inflate threads, reading in `read_size` pieces.  Returns NULL and
// the malloc'd bytes on success, else the error text; counts[0] / counts[1] = members served by workers / sequentially.
char *lash_host_pgzip_read(const char *path, int threads, uint64_t read_size, uint8_t **out, uint64_t *out_bytes, uint64_t *counts)
{
    ParallelGzip pg;
    std::string err = pg.open(path, threads);
    if (!err.empty()) return dup_str(err);
    std::vector<uint8_t> all, buf(read_size ? read_size : 1);
    for (;;) {
        const long r = pg.read(buf.data(), buf.size(), err);
        if (r < 0) return dup_str(err);
        if (r == 0) break;
        all.insert(all.end(), buf.begin(), buf.begin() + r);
    }
    *out_bytes = all.size();
    *out = (uint8_t *)malloc(all.size() + 1);
    if (!all.empty()) memcpy(*out, all.data(), all.size());
    counts[0] = pg.members_parallel();
    counts[1] = pg.members_sequential();
    return nullptr;
}

// chunk-cut rule of the large-file streamer: returns the cut, copies the carry (<= 64 bytes) out
uint64_t lash_host_stream_find_cut(const uint8_t *buf, uint64_t n, int fmt, uint8_t *carry_out, uint64_t *carry_len)
{
    std::vector<uint8_t> carry;
    const size_t cut = stream_find_cut(buf, (size_t)n, fmt, carry);
    *carry_len = carry.size();
    if (!carry.empty()) memcpy(carry_out, carry.data(), std::min<size_t>(carry.size(), 64));
    return cut;
}

char *lash_host_write_parameters(const char *output_name, const char *algorithm, int k, int precision, uint64_t seed)
{
    std::string err = write_parameters_json(output_name, algorithm, k, precision, seed);
    return err.empty() ? nullptr : dup_str(err);
}

// list file -> '\n'-joined kept lines (malloc'd), count in *n
char *lash_host_read_list(const char *path, uint64_t *n)
{
    std::vector<std::string> files;
    std::string err = read_list_file(path, files);
    if (!err.empty()) { *n = (uint64_t)-1; return dup_str(err); }
    std::string j;
    for (auto &f : files) { j += f; j += '\n'; }
    *n = files.size();
    return dup_str(j);
}

char *lash_host_zstd_write(const char *path, const uint8_t *data, uint64_t n, int level, int workers)
{
    ZstdWriter zw;
    std::string err = zw.open(path, level, workers);
    if (err.empty()) err = zw.write(data, (size_t)(n / 2));
    if (err.empty()) err = zw.write(data + n / 2, (size_t)(n - n / 2));
    if (err.empty()) err = zw.finish();
    return err.empty() ? nullptr : dup_str(err);
}

char *lash_host_zstd_read(const char *path, uint8_t **out, uint64_t *n)
{
    std::vector<uint8_t> v;
    std::string err = zstd_decompress_file(path, v);
    if (!err.empty()) return dup_str(err);
    *n = v.size();
    *out = (uint8_t *)malloc(v.size() + 1);
    if (!v.empty()) memcpy(*out, v.data(), v.size());
    return nullptr;
}

// dist_format.hpp for callers outside C++ (lash_amd/allpairs.py): names and cardinalities are handed over once; every block of
// reference rows then goes from the GPU's pair tables to text on `threads` host threads and is written to `fd` in row order.
struct lash_host_formatter {
    std::vector<std::string> row_names, col_names, col_tab;
    std::vector<double> row_card, col_card;
    std::vector<uint32_t> row_id, col_id;
    std::string err;
    lashhost::RowText text;              // the last block's text (memory kept from block to block)
};

// rows and columns in printing order (a pair of equal names prints 0, main.rs:452-453)
lash_host_formatter *lash_host_formatter_create(const char *const *row_names, const double *row_card, uint32_t n_rows,
                                                const char *const *col_names, const double *col_card, uint32_t n_cols)
{
    lash_host_formatter *f = new lash_host_formatter();
    for (uint32_t i = 0; i < n_rows; ++i) f->row_names.emplace_back(row_names[i]);
    for (uint32_t i = 0; i < n_cols; ++i) f->col_names.emplace_back(col_names[i]);
    f->row_card.assign(row_card, row_card + n_rows);
    f->col_card.assign(col_card, col_card + n_cols);
    name_ids(f->row_names, f->col_names, f->row_id, f->col_id);
    f->col_tab = tabbed_names(f->col_names);
    return f;
}

void lash_host_formatter_free(lash_host_formatter *f) { delete f; }
const char *lash_host_formatter_error(const lash_host_formatter *f) { return f ? f->err.c_str() : ""; }

// rows [i0, i1) from their pair tables (row-major, pitch ld; see lash_dist_rows for which a sketch type uses) to `fd`.
// Returns the bytes written, or -1 (lash_host_formatter_error says why).
int64_t lash_host_formatter_block(lash_host_formatter *f, int algo, int p, int k, int model, int fp32, const void *hll_bias, uint32_t i0, uint32_t i1,
                                  int triangle, const uint32_t *c_or_zero, const uint32_t *n_counts, const double *sum_or_union, const double *hmh_ec,
                                  uint64_t ld, int matrix, int threads, int fd)
{
    if (!f || i0 > i1 || i1 > f->row_names.size()) return -1;
    BlockTables t;
    t.c_or_zero = c_or_zero; t.n_counts = n_counts; t.sum_or_union = sum_or_union; t.hmh_ec = hmh_ec; t.ld = ld;
    RowText &text = f->text;                                                     // (one caller at a time: lash_amd/allpairs.py formats blocks in order)
    f->err = dist_block_rows(algo, p, k, model, fp32 != 0, hll_bias, i0, i1, triangle != 0, (uint32_t)f->col_names.size(), f->row_card.data(),
                             f->col_card.data(), t, f->row_names, f->col_names, f->col_tab, f->row_id.data(), f->col_id.data(), matrix != 0, threads, text);
    if (!f->err.empty()) return -1;
    int64_t total = 0;
    for (size_t r = 0; r < text.rows(); ++r) {
        size_t at = 0;
        while (at < text.size(r)) {
            const ssize_t w = write(fd, text.data(r) + at, text.size(r) - at);
            if (w < 0) { f->err = "write failed"; return -1; }
            at += (size_t)w;
        }
        total += (int64_t)text.size(r);
    }
    return total;
}

}  // extern "C"
