// inflate_fast.hpp — a DEFLATE (RFC 1951) decoder for the feed path: the gzip'd FASTA / FASTQ files that `lash sketch` is
// pointed at (the reference reads them through needletail -> flate2, utils.rs:453).  zlib 1.2.11's inflate + crc32, which the
// host readers used until round 2, deliver ~210 MB/s of sequence text per core; the kernels take 600 GB/s, so every `.gz` run
// of the CLI is inflate-bound.  This decoder is written for that data (literal-heavy streams over a tiny alphabet, short
// matches at long distances): 64-bit bit buffer refilled without branches, 11-bit first-level tables, up to three literals per
// refill, 8-byte match copies.  It is an independent implementation of the published format; tests/test_inflate.py checks it
// byte for byte against zlib on streams of every block type and level, and every caller falls back to zlib if it reports an
// error, so a defect here can cost time, never data.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>

namespace lashhost {

class InflateStream {
public:
    enum Status { DONE = 0, OUTPUT_FULL = 1, BAD_DATA = -1, TRUNCATED = -2 };
    // A call makes progress only while out_cap - out_pos >= MIN_ROOM (a whole match plus copy slack must fit).
    static constexpr size_t MIN_ROOM = 320;

    InflateStream() { begin(); }
    void begin();                        // start of a raw deflate stream
    // Decodes from in[in_pos ..) into out[out_pos .. out_cap).  out[0 .. out_pos) must hold what this stream has produced so
    // far — all of it, or (after the caller slid its window) at least the last 32 KiB.  The same `in` buffer must be passed
    // to every call of one stream.  DONE: the final block ended, in_pos = first byte after the stream (the gzip trailer).
    Status run(const uint8_t *in, size_t in_n, size_t &in_pos, uint8_t *out, size_t &out_pos, size_t out_cap);

private:
    enum Phase { PH_HEADER, PH_STORED, PH_HUFF, PH_DONE };
#ifndef LASH_INFLATE_LL_ROOT
#define LASH_INFLATE_LL_ROOT 11          // first-level bits of the literal/length table (tools/inflate_root_scan.sh measures 9..12)
#endif
    static constexpr int LL_ROOT = LASH_INFLATE_LL_ROOT, D_ROOT = 8;
    static constexpr int LL_CAP = (1 << LL_ROOT) + 1024, D_CAP = (1 << D_ROOT) + 512;
    template <bool FAST>
    int decode_block(const uint8_t *in, size_t in_n, size_t &ip, uint8_t *out, size_t &op, size_t out_cap);
    bool read_dynamic_header(const uint8_t *in, size_t in_n, size_t &ip, int &err);
    bool need_bits(const uint8_t *in, size_t in_n, size_t &ip, unsigned n);

    uint64_t bitbuf_;
    unsigned bitcnt_;
    Phase phase_;
    bool final_;
    size_t stored_left_;
    void build_literal_runs();
    uint32_t ll_[LL_CAP];
    uint32_t d_[D_CAP];
    uint32_t ml_[1 << LL_ROOT];          // per 11-bit window: up to 3 literals decoded at once (see build_literal_runs)
};

// CRC-32 (IEEE 802.3, the gzip trailer's): carry-less-multiply folding where the CPU has PCLMULQDQ (>10 GB/s), else slicing-by-8.
uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n);

// Every member of the gzip file image src[0..n): header (RFC 1952, all optional fields), deflate stream, CRC-32 and length
// check; the inflated bytes are appended to out.  Returns nullptr on success, else a static string saying what is wrong.  `one_member`: stop after the first member and
// report where it ended through *consumed.
struct ByteSink {                      // contiguous growable output (realloc; the decoder only keeps offsets)
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    size_t limit = ~(size_t)0;         // give up (error "too large") beyond this many bytes
    ~ByteSink();
    ByteSink() = default;
    ByteSink(const ByteSink &) = delete;
    ByteSink &operator=(const ByteSink &) = delete;
    bool reserve(size_t want);
    void release();
    void swap(ByteSink &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(limit, o.limit); }
};
const char *gunzip_members(const uint8_t *src, size_t n, ByteSink &out, bool one_member, size_t *consumed);

// Length of the RFC 1952 member header at p (>= 10), or 0: not a gzip member, unsupported method / flags, or cut short.
size_t gzip_header_length(const uint8_t *p, size_t avail);

// One gzip member's deflate body read sequentially with bounded memory (a single-member file of any size): 32 KiB of
// history + `chunk` bytes, refilled as the caller drains it.  The caller positions in_pos after the member header and, when
// done() turns true, finds it at the member's 8-byte trailer, which it checks against crc() / total().  crc() / total() always
// cover exactly the bytes read() has handed out so far.  TRUNCATED is terminal: `in` must be the complete stream (the callers
// pass the whole mmap'd file); the decoder consumes bits before it reports it and cannot be resumed.
class WindowedInflate {
public:
    explicit WindowedInflate(size_t chunk = 1u << 20);
    ~WindowedInflate();
    WindowedInflate(const WindowedInflate &) = delete;
    WindowedInflate &operator=(const WindowedInflate &) = delete;
    void begin();
    // up to n bytes into dst; returns how many (0 only when done()), or -1 with *err set
    long read(const uint8_t *in, size_t in_n, size_t &in_pos, uint8_t *dst, size_t n, const char **err);
    bool done() const { return done_ && rd_ == op_; }
    uint32_t crc() const { return crc_; }
    uint64_t total() const { return total_; }
    void test_flip_output_byte(long at) { test_flip_ = at; }   // tests: a decoder defect — output byte `at` of every member comes out wrong
private:
    InflateStream z_;
    uint8_t *buf_;
    size_t cap_, op_ = 0, rd_ = 0;
    bool done_ = false;
    uint32_t crc_ = 0;
    uint64_t total_ = 0;
    long test_flip_ = -1;
};

}  // namespace lashhost
