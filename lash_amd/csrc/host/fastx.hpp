// fastx.hpp — FASTA/FASTQ record reader for the host side of lash-gfx950 (replaces needletail::parse_fastx_file,
// /root/reference/src/utils.rs:453-459; semantics per SURVEY.md App. A.5).
//   * compression is sniffed from magic bytes like needletail does: gzip (zlib), bzip2 and xz (dlopen'd libbz2 /
//     liblzma, codec_dl.hpp), zstd (dlopen'd libzstd), plain;
//   * '>' => FASTA: header line, then sequence lines up to the next '>' with '\n' and '\r' stripped;
//     '@' => FASTQ: 4-line records, the sequence is line 2;
//   * seq() bytes are appended UNFILTERED: deleting non-ACGT bytes is the pack kernel's job (utils.rs:459 -> 33-41).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lashhost {

struct RecordBatch {
    std::vector<uint8_t> seq;          // concatenated record sequences
    std::vector<uint64_t> rec_off;     // n_rec + 1 offsets into seq (rec_off[0] == 0 for a fresh batch)
    RecordBatch() { rec_off.push_back(0); }
    uint64_t n_rec() const { return rec_off.size() - 1; }
};

// Appends every record of `path` to `out`.  Returns "" on success, else an error message
// (the reference panics with "Invalid input file", utils.rs:453).
std::string read_fastx_file(const std::string &path, RecordBatch &out);

// Parses an in-memory (already decompressed) FASTA/FASTQ buffer.
std::string parse_fastx_buffer(const uint8_t *data, size_t n, RecordBatch &out);

// Sequential reader of a file's uncompressed bytes (plain, gzip via zlib, bzip2 / xz / zstd via dlopen'd libraries): the
// large-file path streams chunks through it instead of holding a whole metagenome in memory.
class ByteStream {
public:
    ByteStream();
    ~ByteStream();
    // > 1: a gzip file is read by pgzip.hpp's multi-member reader with that many inflate threads (call before open())
    void set_threads(int threads);
    std::string open(const std::string &path);
    // reads up to n bytes; returns the count (0 = end of data) or -1 with err set
    long read(uint8_t *dst, size_t n, std::string &err);
private:
    struct Impl;
    Impl *impl_;
};

// Whole file into memory, transparently inflating gzip / bzip2 / xz / zstd.
std::string slurp_maybe_compressed(const std::string &path, std::vector<uint8_t> &out);

}  // namespace lashhost
