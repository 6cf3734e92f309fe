// pgzip.hpp — parallel reader of MULTI-MEMBER gzip files (concatenated members: bgzip, pigz -i, `cat a.gz b.gz`, sequencer
// lane files glued together).  A gzip stream cannot be entered in the middle, but every member can be inflated on its own:
// candidate member starts are found by their magic bytes, speculative workers inflate from the candidates ahead of the
// read position, and a result is used only if the previous member ENDED exactly where it starts (zlib checks each member's
// CRC-32 / length trailer), so a false candidate — the magic inside compressed data — costs a wasted worker, never a wrong
// byte.  A single-member file, a member larger than the per-member buffer, or a member whose header the scan does not
// recognise is inflated sequentially from the read position, exactly like a plain gzip reader.
//
// Members are inflated by inflate_fast.hpp's decoder (1.5-1.7x zlib on sequence text); whatever it rejects is taken
// sequentially, where zlib has the last word.  Bytes after a complete member that do not start another member (zero padding)
// end the data, with any number of threads.
//
// Feed path of BASELINE configs[4] (one huge FASTQ.gz): ~0.5-0.9 GB/s of text per core, the kernels take 600 GB/s.
#pragma once
#include <cstdint>
#include <string>

namespace lashhost {

namespace test_seams { extern long inflate_fail_after, inflate_flip_at; }   // see pgzip.cpp; set only by host_hooks.cpp

class ParallelGzip {
public:
    ParallelGzip();
    ~ParallelGzip();
    // threads <= 1: plain sequential inflate of the mapped file
    std::string open(const std::string &path, int threads);
    long read(uint8_t *dst, size_t n, std::string &err);   // bytes (0 = end of data) or -1 with err set
    uint64_t members_parallel() const;                     // members served from a speculative worker (tests / stats)
    uint64_t members_sequential() const;
private:
    struct Impl;
    Impl *impl_;
};

}  // namespace lashhost
