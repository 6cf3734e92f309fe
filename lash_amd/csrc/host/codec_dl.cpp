#include "codec_dl.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace lashhost {
namespace {

// ---- libbz2 1.0 (bzlib.h) ----
struct bz_stream {
    char *next_in;
    unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out;
    unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int);
    void (*bzfree)(void *, void *);
    void *opaque;
};
constexpr int BZ_OK = 0, BZ_STREAM_END = 4;
using bz_init_t = int (*)(bz_stream *, int, int);
using bz_run_t = int (*)(bz_stream *);

// ---- liblzma 5.x (lzma/base.h) ----
struct lzma_stream {
    const uint8_t *next_in;
    size_t avail_in;
    uint64_t total_in;
    uint8_t *next_out;
    size_t avail_out;
    uint64_t total_out;
    const void *allocator;
    void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2;
    size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
constexpr int LZMA_OK = 0, LZMA_STREAM_END = 1, LZMA_RUN = 0, LZMA_FINISH = 3;
constexpr uint32_t LZMA_CONCATENATED = 0x08;
using lzma_dec_t = int (*)(lzma_stream *, uint64_t, uint32_t);
using lzma_code_t = int (*)(lzma_stream *, int);
using lzma_end_t = void (*)(lzma_stream *);

struct Libs {
    bz_init_t bz_init = nullptr;
    bz_run_t bz_run = nullptr, bz_end = nullptr;
    lzma_dec_t xz_dec = nullptr;
    lzma_code_t xz_code = nullptr;
    lzma_end_t xz_end = nullptr;
    std::string bz_err, xz_err;
};

Libs &libs()
{
    static Libs L;
    static std::once_flag once;
    std::call_once(once, [] {
        if (void *h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL)) {
            L.bz_init = (bz_init_t)dlsym(h, "BZ2_bzDecompressInit");
            L.bz_run = (bz_run_t)dlsym(h, "BZ2_bzDecompress");
            L.bz_end = (bz_run_t)dlsym(h, "BZ2_bzDecompressEnd");
            if (!L.bz_init || !L.bz_run || !L.bz_end) L.bz_err = "libbz2.so.1 lacks the BZ2_bzDecompress* symbols";
        } else L.bz_err = "libbz2.so.1 not found";
        if (void *h = dlopen("liblzma.so.5", RTLD_NOW | RTLD_LOCAL)) {
            L.xz_dec = (lzma_dec_t)dlsym(h, "lzma_stream_decoder");
            L.xz_code = (lzma_code_t)dlsym(h, "lzma_code");
            L.xz_end = (lzma_end_t)dlsym(h, "lzma_end");
            if (!L.xz_dec || !L.xz_code || !L.xz_end) L.xz_err = "liblzma.so.5 lacks the lzma_* decoder symbols";
        } else L.xz_err = "liblzma.so.5 not found";
    });
    return L;
}

}  // namespace

bool codec_available(Codec codec, std::string *why)
{
    Libs &L = libs();
    const std::string &e = codec == Codec::BZIP2 ? L.bz_err : L.xz_err;
    if (why) *why = e;
    return e.empty();
}

struct DlDecoder::Impl {
    Codec codec = Codec::BZIP2;
    FILE *f = nullptr;
    bz_stream bz{};
    lzma_stream xz{};
    bool bz_live = false, xz_live = false, eof_in = false, done = false;
    std::vector<uint8_t> in;
    size_t in_pos = 0, in_size = 0;
    bool refill()
    {
        if (eof_in) return false;
        in_size = fread(in.data(), 1, in.size(), f);
        in_pos = 0;
        if (in_size == 0) { eof_in = true; return false; }
        return true;
    }
};

DlDecoder::DlDecoder() : impl_(new Impl()) {}
DlDecoder::~DlDecoder()
{
    Libs &L = libs();
    if (impl_->bz_live) L.bz_end(&impl_->bz);
    if (impl_->xz_live) L.xz_end(&impl_->xz);
    if (impl_->f) fclose(impl_->f);
    delete impl_;
}

std::string DlDecoder::open(FILE *f, Codec codec)
{
    impl_->f = f;
    impl_->codec = codec;
    impl_->in.resize(1 << 20);
    std::string why;
    if (!codec_available(codec, &why))
        return std::string("Invalid input file: ") + (codec == Codec::BZIP2 ? "bzip2" : "xz") + " input needs " + why;
    Libs &L = libs();
    if (codec == Codec::BZIP2) {
        if (L.bz_init(&impl_->bz, 0, 0) != BZ_OK) return "Invalid input file: BZ2_bzDecompressInit failed";
        impl_->bz_live = true;
    } else {
        if (L.xz_dec(&impl_->xz, UINT64_MAX, LZMA_CONCATENATED) != LZMA_OK) return "Invalid input file: lzma_stream_decoder failed";
        impl_->xz_live = true;
    }
    return "";
}

long DlDecoder::read(uint8_t *dst, size_t n, std::string &err)
{
    Impl &s = *impl_;
    Libs &L = libs();
    size_t done = 0;
    while (done < n && !s.done) {
        if (s.in_pos == s.in_size) s.refill();
        const size_t want = std::min<size_t>(n - done, 1u << 30);
        if (s.codec == Codec::BZIP2) {
            if (s.in_pos == s.in_size && s.eof_in) {                 // input exhausted between streams: clean end
                if (!s.bz_live) { s.done = true; break; }
                err = "Invalid input file: truncated bzip2 stream";
                return -1;
            }
            if (!s.bz_live) {                                        // another concatenated stream follows
                memset(&s.bz, 0, sizeof s.bz);
                if (L.bz_init(&s.bz, 0, 0) != BZ_OK) { err = "Invalid input file: BZ2_bzDecompressInit failed"; return -1; }
                s.bz_live = true;
            }
            s.bz.next_in = reinterpret_cast<char *>(s.in.data() + s.in_pos);
            s.bz.avail_in = (unsigned)(s.in_size - s.in_pos);
            s.bz.next_out = reinterpret_cast<char *>(dst + done);
            s.bz.avail_out = (unsigned)want;
            const int rc = L.bz_run(&s.bz);
            s.in_pos = s.in_size - s.bz.avail_in;
            done += want - s.bz.avail_out;
            if (rc == BZ_STREAM_END) { L.bz_end(&s.bz); s.bz_live = false; }
            else if (rc != BZ_OK) { err = "Invalid input file: corrupt bzip2 stream"; return -1; }
        } else {
            s.xz.next_in = s.in.data() + s.in_pos;
            s.xz.avail_in = s.in_size - s.in_pos;
            s.xz.next_out = dst + done;
            s.xz.avail_out = want;
            const int rc = L.xz_code(&s.xz, (s.eof_in && s.xz.avail_in == 0) ? LZMA_FINISH : LZMA_RUN);
            s.in_pos = s.in_size - s.xz.avail_in;
            done += want - s.xz.avail_out;
            if (rc == LZMA_STREAM_END) { s.done = true; break; }
            if (rc != LZMA_OK) { err = "Invalid input file: corrupt xz stream"; return -1; }
        }
    }
    return (long)done;
}

}  // namespace lashhost
