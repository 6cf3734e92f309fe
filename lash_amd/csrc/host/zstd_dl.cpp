#include "zstd_dl.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace lashhost {
namespace {

struct InBuf { const void *src; size_t size; size_t pos; };
struct OutBuf { void *dst; size_t size; size_t pos; };

struct Api {
    void *h = nullptr;
    void *(*createCStream)() = nullptr;
    size_t (*freeCStream)(void *) = nullptr;
    size_t (*initCStream)(void *, int) = nullptr;
    size_t (*compressStream)(void *, OutBuf *, InBuf *) = nullptr;
    size_t (*endStream)(void *, OutBuf *) = nullptr;
    size_t (*CCtx_setParameter)(void *, int, int) = nullptr;
    size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
    size_t (*compressBound)(size_t) = nullptr;
    void *(*createDStream)() = nullptr;
    size_t (*freeDStream)(void *) = nullptr;
    size_t (*initDStream)(void *) = nullptr;
    size_t (*decompressStream)(void *, OutBuf *, InBuf *) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char *(*getErrorName)(size_t) = nullptr;
    std::string why;
};

Api &api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"libzstd.so.1", "libzstd.so", "/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1"};
        for (const char *n : names) {
            a.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (a.h) break;
        }
        if (!a.h) { a.why = "libzstd.so.1 not found"; return; }
        auto sym = [&](const char *s) { return dlsym(a.h, s); };
        a.createCStream = (void *(*)())sym("ZSTD_createCStream");
        a.freeCStream = (size_t(*)(void *))sym("ZSTD_freeCStream");
        a.initCStream = (size_t(*)(void *, int))sym("ZSTD_initCStream");
        a.compressStream = (size_t(*)(void *, OutBuf *, InBuf *))sym("ZSTD_compressStream");
        a.endStream = (size_t(*)(void *, OutBuf *))sym("ZSTD_endStream");
        a.CCtx_setParameter = (size_t(*)(void *, int, int))sym("ZSTD_CCtx_setParameter");
        a.compress = (size_t(*)(void *, size_t, const void *, size_t, int))sym("ZSTD_compress");
        a.compressBound = (size_t(*)(size_t))sym("ZSTD_compressBound");
        a.createDStream = (void *(*)())sym("ZSTD_createDStream");
        a.freeDStream = (size_t(*)(void *))sym("ZSTD_freeDStream");
        a.initDStream = (size_t(*)(void *))sym("ZSTD_initDStream");
        a.decompressStream = (size_t(*)(void *, OutBuf *, InBuf *))sym("ZSTD_decompressStream");
        a.isError = (unsigned (*)(size_t))sym("ZSTD_isError");
        a.getErrorName = (const char *(*)(size_t))sym("ZSTD_getErrorName");
        if (!a.createCStream || !a.initCStream || !a.compressStream || !a.endStream || !a.createDStream ||
            !a.decompressStream || !a.isError || !a.freeCStream || !a.freeDStream || !a.initDStream) {
            a.why = "libzstd lacks the streaming API";
            a.h = nullptr;
        }
    });
    return a;
}

std::string zerr(size_t code)
{
    Api &a = api();
    return std::string("zstd: ") + (a.getErrorName ? a.getErrorName(code) : "error");
}

}  // namespace

bool zstd_available(std::string *why)
{
    Api &a = api();
    if (!a.h && why) *why = a.why;
    return a.h != nullptr;
}

ZstdWriter::ZstdWriter() {}
ZstdWriter::~ZstdWriter()
{
    if (cstream_) api().freeCStream(cstream_);
    if (f_) fclose(f_);
}

std::string ZstdWriter::open(const std::string &path, int level, int workers)
{
    Api &a = api();
    if (!a.h) return "zstd unavailable: " + a.why;
    f_ = fopen(path.c_str(), "wb");
    if (!f_) return "cannot create " + path;
    cstream_ = a.createCStream();
    if (!cstream_) return "ZSTD_createCStream failed";
    size_t rc = a.initCStream(cstream_, level);
    if (a.isError(rc)) return zerr(rc);
    // ZSTD_c_nbWorkers = 400; an error when the library was built without multithreading: then frames from our own threads
    level_ = level;
    workers_ = std::max(1, std::min(workers, 16));          // (a round of the threads is workers x 4 MiB of images: 64 MiB at most)
    bool mt = false;
    if (workers > 1 && a.CCtx_setParameter) mt = !a.isError(a.CCtx_setParameter(cstream_, 400, workers));
    if (getenv("LASH_ZSTD_ONE_THREAD")) { mt = true; workers_ = 1; }    // (A/B and tests: the plain single-frame path)
    frames_ = !mt && workers_ > 1 && a.compress && a.compressBound;
    out_.resize(1 << 20);
    return "";
}

// p[0 .. (n_chunks - 1) * kChunk + last_bytes) as n_chunks frames, compressed by up to workers_ threads, written in order
std::string ZstdWriter::frames(const uint8_t *p, size_t n_chunks, size_t last_bytes)
{
    Api &a = api();
    std::vector<std::vector<uint8_t>> outs(n_chunks);
    std::vector<size_t> sizes(n_chunks, 0);
    std::string err;
    std::mutex em;
    const size_t T = std::min<size_t>((size_t)workers_, n_chunks);
    auto work = [&](size_t t) {
        for (size_t i = t; i < n_chunks; i += T) {
            const size_t n = i + 1 == n_chunks ? last_bytes : kChunk;
            outs[i].resize(a.compressBound(n));
            const size_t rc = a.compress(outs[i].data(), outs[i].size(), p + i * kChunk, n, level_);
            if (a.isError(rc)) { std::lock_guard<std::mutex> g(em); if (err.empty()) err = zerr(rc); return; }
            sizes[i] = rc;
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (!err.empty()) return err;
    for (size_t i = 0; i < n_chunks; ++i)
        if (fwrite(outs[i].data(), 1, sizes[i], f_) != sizes[i]) return "short write";
    wrote_frame_ = wrote_frame_ || n_chunks > 0;
    return "";
}

std::string ZstdWriter::write(const void *data, size_t n)
{
    Api &a = api();
    if (frames_) {
        const uint8_t *p = static_cast<const uint8_t *>(data);
        const size_t round = (size_t)workers_ * kChunk;                 // one round of the threads
        if (!pend_.empty() || n < round) {
            // small writes (a batch of a dozen 5 Mbp genomes is 400 KB of images) gather until every thread has a piece
            const size_t take = std::min(n, round - std::min(round, pend_.size()));
            pend_.insert(pend_.end(), p, p + take);
            p += take; n -= take;
            if (pend_.size() < round) return "";
            std::string e = frames(pend_.data(), pend_.size() / kChunk, kChunk);
            pend_.clear();
            if (!e.empty()) return e;
        }
        // large writes straight from the caller's buffer, a ROUND of the threads at a time: frames() holds every chunk's compressed bytes
        // until all threads have joined, so one call over a whole multi-GB write would take memory of the write's size, not the
        // workers x 4 MiB the class promises (a batch of thousands of HyperMinHash images is hundreds of MB per write)
        const size_t full = n / kChunk;
        for (size_t done = 0; done < full; done += (size_t)workers_) {
            std::string e = frames(p + done * kChunk, std::min<size_t>((size_t)workers_, full - done), kChunk);
            if (!e.empty()) return e;
        }
        pend_.insert(pend_.end(), p + full * kChunk, p + n);
        return "";
    }
    InBuf in{data, n, 0};
    while (in.pos < in.size) {
        OutBuf ob{out_.data(), out_.size(), 0};
        size_t rc = a.compressStream(cstream_, &ob, &in);
        if (a.isError(rc)) return zerr(rc);
        if (ob.pos && fwrite(out_.data(), 1, ob.pos, f_) != ob.pos) return "short write";
    }
    return "";
}

std::string ZstdWriter::finish()
{
    Api &a = api();
    if (frames_) {
        // what is left — and for an empty stream ONE empty frame: the file must be a zstd stream
        if (!pend_.empty() || !wrote_frame_) {
            const size_t nc = std::max<size_t>(1, (pend_.size() + kChunk - 1) / kChunk);
            const size_t last = pend_.size() - (nc - 1) * kChunk;
            std::string e = frames(pend_.data(), nc, last);
            pend_.clear();
            if (!e.empty()) return e;
        }
        a.freeCStream(cstream_);
        cstream_ = nullptr;
        int e = fclose(f_);
        f_ = nullptr;
        return e ? "close failed" : "";
    }
    for (;;) {
        OutBuf ob{out_.data(), out_.size(), 0};
        size_t rc = a.endStream(cstream_, &ob);
        if (a.isError(rc)) return zerr(rc);
        if (ob.pos && fwrite(out_.data(), 1, ob.pos, f_) != ob.pos) return "short write";
        if (rc == 0) break;
    }
    a.freeCStream(cstream_);
    cstream_ = nullptr;
    int e = fclose(f_);
    f_ = nullptr;
    return e ? "close failed" : "";
}

ZstdReader::ZstdReader() {}
ZstdReader::~ZstdReader()
{
    if (ds_) api().freeDStream(ds_);
    if (f_) fclose(f_);
}

std::string ZstdReader::open(FILE *f)
{
    f_ = f;
    Api &a = api();
    if (!a.h) return "zstd unavailable: " + a.why;
    ds_ = a.createDStream();
    if (!ds_) return "ZSTD_createDStream failed";
    a.initDStream(ds_);
    in_.resize(1 << 20);
    return "";
}

long ZstdReader::read(uint8_t *dst, size_t n, std::string &err)
{
    Api &a = api();
    OutBuf ob{dst, n, 0};
    while (ob.pos < ob.size) {
        if (in_pos_ == in_size_ && !eof_) {
            in_size_ = fread(in_.data(), 1, in_.size(), f_);
            in_pos_ = 0;
            if (in_size_ == 0) eof_ = true;
        }
        if (in_pos_ == in_size_ && eof_) break;
        InBuf ib{in_.data(), in_size_, in_pos_};
        size_t rc = a.decompressStream(ds_, &ob, &ib);
        in_pos_ = ib.pos;
        if (a.isError(rc)) { err = zerr(rc); return -1; }
        if (rc == 0 && (in_pos_ < in_size_ || !eof_)) a.initDStream(ds_);      // next frame, if any
    }
    return (long)ob.pos;
}

std::string zstd_decompress_all(const uint8_t *src, size_t n, std::vector<uint8_t> &out)
{
    Api &a = api();
    if (!a.h) return "zstd unavailable: " + a.why;
    void *ds = a.createDStream();
    if (!ds) return "ZSTD_createDStream failed";
    a.initDStream(ds);
    out.clear();
    std::vector<uint8_t> buf(1 << 22);
    InBuf in{src, n, 0};
    size_t rc = 1;
    while (in.pos < in.size) {
        OutBuf ob{buf.data(), buf.size(), 0};
        rc = a.decompressStream(ds, &ob, &in);
        if (a.isError(rc)) { a.freeDStream(ds); return zerr(rc); }
        out.insert(out.end(), buf.begin(), buf.begin() + ob.pos);
        if (rc == 0 && in.pos < in.size) a.initDStream(ds);        // concatenated frames
    }
    a.freeDStream(ds);
    return "";
}

std::string zstd_decompress_file(const std::string &path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return "Error opening " + path;
    std::vector<uint8_t> raw, buf(1 << 22);
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) raw.insert(raw.end(), buf.begin(), buf.begin() + n);
    fclose(f);
    return zstd_decompress_all(raw.data(), raw.size(), out);
}

}  // namespace lashhost
