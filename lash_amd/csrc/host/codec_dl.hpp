// codec_dl.hpp — bzip2 and xz stream decoders through dlopen("libbz2.so.1") / dlopen("liblzma.so.5").
// needletail's default features sniff and inflate gzip, bzip2 and xz inputs (parse_fastx_file,
// /root/reference/src/utils.rs:453); the image ships both runtime libraries without headers, so the few prototypes and
// the two stream structs of their stable C ABIs are declared in codec_dl.cpp.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace lashhost {

enum class Codec { BZIP2, XZ };

// incremental decoder over a FILE* (for ByteStream); concatenated streams are decoded back to back
class DlDecoder {
public:
    DlDecoder();
    ~DlDecoder();
    std::string open(FILE *f, Codec codec);     // takes ownership of f
    long read(uint8_t *dst, size_t n, std::string &err);
private:
    struct Impl;
    Impl *impl_;
};

bool codec_available(Codec codec, std::string *why = nullptr);

}  // namespace lashhost
