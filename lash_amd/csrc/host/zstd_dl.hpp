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

// Streaming compressor writing to a FILE*: one zstd frame, level `level`, `workers` zstdmt workers if supported.
class ZstdWriter {
public:
    ZstdWriter();
    ~ZstdWriter();
    std::string open(const std::string &path, int level, int workers);
    std::string write(const void *data, size_t n);
    std::string finish();                       // ends the frame and closes the file
private:
    void *cstream_ = nullptr;
    FILE *f_ = nullptr;
    std::vector<uint8_t> out_;
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
