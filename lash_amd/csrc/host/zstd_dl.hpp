// zstd_dl.hpp — the zstd streaming API through dlopen("libzstd.so.1") (the image ships the library without its
// header).  Replaces the `zstd` crate calls of the reference: Encoder::new(w, 3) + multithread(threads) + finish()
// (/root/reference/src/utils.rs:567-574) and Decoder::new (utils.rs:98,211,312).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace lashhost {

bool zstd_available(std::string *why = nullptr);

// Streaming compressor writing to a FILE*, level `level`.  With a libzstd built for multithreading: ONE frame, `workers` zstdmt workers
// (what the reference's `encoder.multithread(threads)` does, utils.rs:568-569).  With one that is not (this image's 1.4.8 refuses
// ZSTD_c_nbWorkers): the stream is cut into 4 MiB pieces which `workers` threads compress as independent FRAMES, written in order — a
// concatenation of frames is a valid zstd stream (RFC 8878 section 3; the `zstd` crate's Decoder reads on into the next frame unless
// `single_frame()` is asked for), the decompressed bytes are the same.  100 000 HyperMinHash images (3.3 GB) took 14 s on one thread.
class ZstdWriter {
public:
    ZstdWriter();
    ~ZstdWriter();
    std::string open(const std::string &path, int level, int workers);
    std::string write(const void *data, size_t n);
    std::string finish();                       // ends the frame and closes the file
private:
    std::string frames(const uint8_t *p, size_t n_chunks, size_t last_bytes);   // chunks of kChunk bytes (the last: last_bytes) -> frames
    static constexpr size_t kChunk = 4u << 20;
    void *cstream_ = nullptr;
    FILE *f_ = nullptr;
    std::vector<uint8_t> out_;
    bool frames_ = false;                       // independent frames from `workers_` threads
    int workers_ = 1, level_ = 3;
    bool wrote_frame_ = false;
    std::vector<uint8_t> pend_;                 // frames_: bytes not yet compressed (< workers_ * kChunk)
};

// incremental decoder over a FILE* (for ByteStream)
class ZstdReader {
public:
    ZstdReader();
    ~ZstdReader();
    std::string open(FILE *f);                 // takes ownership of f
    long read(uint8_t *dst, size_t n, std::string &err);
private:
    void *ds_ = nullptr;
    FILE *f_ = nullptr;
    std::vector<uint8_t> in_;
    size_t in_pos_ = 0, in_size_ = 0;
    bool eof_ = false;
};

std::string zstd_decompress_all(const uint8_t *src, size_t n, std::vector<uint8_t> &out);
std::string zstd_decompress_file(const std::string &path, std::vector<uint8_t> &out);

}  // namespace lashhost
