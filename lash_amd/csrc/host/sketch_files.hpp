// sketch_files.hpp — host driver that mirrors the reference's
//     sketch_files::<S>(precision, files, kmer_length, output_name, threads, seed, aa)
// (/root/reference/src/utils.rs:439-583) on top of the C ABI: files are parsed by a pool of reader threads,
// grouped into batches in file order, sketched on one or more GPUs (one lash_ctx per device, one host thread each,
// contiguous batches => output order == file order, utils.rs:509), and the images are streamed through zstd
// level 3 into {output_name}_sketches.bin (utils.rs:567-574); {output_name}_files.json follows (utils.rs:577-580).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/lash_gfx950.h"

namespace lashhost {

// --layout SPEC, else $LASH_LAYOUT, else the default (include/lash_gfx950.h `lash_layout`); "" on success
std::string layout_from_option(const std::string &spec, lash_layout &out);

struct SketchOptions {
    int algo = 0;                    // LASH_HMH / LASH_HLL / LASH_ULL
    int precision = 10;              // -p, ignored for hmh (main.rs:212-213)
    int k = 16;
    uint64_t seed = 42;
    int threads = 1;                 // -t: reader threads and zstd workers (main.rs:184-192)
    std::vector<int> devices;        // GPUs to use; empty = device 0
    uint64_t batch_bytes = 1ull << 26;   // file bytes per GPU batch (page-locking a buffer costs ~0.18 s per GB)
    uint64_t stream_bytes = 1ull << 30;  // files larger than this (compressed: > 1/3 of it on disk) are streamed in chunks
                                         // of this size with on-device accumulation (BASELINE configs[4]); < 4 GiB
    uint32_t flags = 0;              // LASH_F_HMH_X_LOW, LASH_F_AMINO (--aa)
    lash_layout layout;              // set by layout_from_option(); every context gets it
    SketchOptions() { lash_layout_default(&layout); }
};

struct SketchStats {
    uint64_t files = 0, records = 0, bytes = 0, batches = 0;
    double seconds = 0;
};

// Returns "" on success, otherwise the error text (the reference panics / returns Err).
std::string sketch_files(const SketchOptions &opt, const std::vector<std::string> &files,
                         const std::string &output_name, SketchStats *stats = nullptr);

// main.rs:200-207: one path per line, lines that are blank after trim() are dropped, others kept verbatim
std::string read_list_file(const std::string &path, std::vector<std::string> &files);

// main.rs:248-276
std::string write_parameters_json(const std::string &output_name, const std::string &algorithm, int k, int precision,
                                  uint64_t seed, bool amino = false);

// where a streamed chunk ends (0 = no boundary found) and the bytes the next chunk must start with (tests)
size_t stream_find_cut(const uint8_t *b, size_t n, int fmt, std::vector<uint8_t> &carry);

}  // namespace lashhost
