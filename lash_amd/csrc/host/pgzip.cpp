#include "pgzip.hpp"

#include "inflate_fast.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace lashhost {

// fault injectors of the fast-inflate fallback tests (tests/test_host.py): hand the member over to zlib after that many delivered bytes / flip
// that output byte.  Plain variables here, -1 = off; the environment variables that set them (LASH_TEST_FAST_INFLATE_*) are read by a static
// initialiser in host_hooks.cpp, which is linked into liblash_host.so (the tests' library) and NOT into the `lash` command
namespace test_seams { long inflate_fail_after = -1, inflate_flip_at = -1; }

namespace {

size_t member_cap()                                // a speculative worker gives up on a member that inflates beyond this
{
    static const size_t cap = getenv("LASH_PGZIP_MEMBER_CAP") ? (size_t)strtoull(getenv("LASH_PGZIP_MEMBER_CAP"), nullptr, 10) : (512ull << 20);
    return cap;
}
constexpr size_t BUFFER_CAP = 4ull << 30;          // inflated bytes held ahead of the reader, all workers together
constexpr size_t SCAN_AHEAD = 1ull << 30;          // compressed bytes ahead of the read position scanned for candidates

struct Member {
    uint64_t start = 0, end = 0;                   // compressed offsets [start, end)
    ByteSink data;                                 // the member's inflated bytes, contiguous
    int state = 0;                                 // 0 queued, 1 running, 2 done ok, 3 failed / too big
};

// inflate ONE gzip member starting at src[0] (inflate_fast.hpp: header, deflate body, CRC-32 and length); returns the
// consumed bytes, 0 on any failure — a false candidate, a member beyond `cap`, or a stream the fast decoder rejects: the
// reader then takes the member sequentially, where zlib has the last word
size_t inflate_member_zlib(const uint8_t *src, size_t avail, ByteSink &out, size_t cap)   // LASH_NO_FAST_INFLATE=1: zlib only
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return 0;
    size_t used = 0;
    int rc = Z_OK;
    out.n = 0;
    while (rc != Z_STREAM_END) {
        if (z.avail_in == 0) {
            const size_t chunk = std::min<size_t>(avail - used, 1u << 30);
            if (chunk == 0) break;                     // ran out of file before the member ended
            z.next_in = const_cast<Bytef *>(src + used);
            z.avail_in = (uInt)chunk;
            used += chunk;
        }
        if (out.cap - out.n < (1u << 16)) {
            if (out.n > cap || !out.reserve(out.cap + out.cap / 2 + (4u << 20))) { rc = Z_MEM_ERROR; break; }
        }
        z.next_out = out.p + out.n;
        z.avail_out = (uInt)std::min<size_t>(out.cap - out.n, 1u << 30);
        const uInt before = z.avail_out;
        rc = inflate(&z, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) break;
        out.n += before - z.avail_out;
    }
    const size_t consumed = used - z.avail_in;
    inflateEnd(&z);
    return rc == Z_STREAM_END && out.n <= cap ? consumed : 0;
}

size_t inflate_member(const uint8_t *src, size_t avail, ByteSink &out, size_t cap)
{
    static const bool no_fast = getenv("LASH_NO_FAST_INFLATE") != nullptr;
    if (no_fast) {
        const size_t used = inflate_member_zlib(src, avail, out, cap);
        if (!used) out.n = 0;
        return used;
    }
    out.limit = cap;
    out.n = 0;
    size_t used = 0;
    const char *e = gunzip_members(src, avail, out, true, &used);
    if (e) { out.n = 0; return 0; }
    return used;
}

}  // namespace

struct ParallelGzip::Impl {
    const uint8_t *map = nullptr;
    size_t size = 0;
    int fd = -1;
    int threads = 1;
    // sequential decoder state (the member at the read position, when no speculative result serves it)
    z_stream z;
    bool z_open = false;
    uint64_t pos = 0;                              // compressed offset of the member at the read position (always a member boundary)
    uint64_t in_pos = 0;                           // how far the sequential decoder has been fed (>= pos while z_open)
    // ... first tried with the bounded-memory fast decoder (inflate_fast.hpp); zlib takes the member over from its start,
    // skipping what was already handed out, if that one reports an error or a trailer mismatch
    WindowedInflate fast{4u << 20};
    bool fast_open = false;
    size_t fast_ip = 0;
    uint64_t delivered = 0, z_skip = 0;
    const bool no_fast = getenv("LASH_NO_FAST_INFLATE") != nullptr;
    const long test_fail_after = test_seams::inflate_fail_after;   // (-1 in the product: only liblash_host.so's hooks ever set these, host_hooks.cpp)
    const long test_flip_at = test_seams::inflate_flip_at;         // corrupt that output byte
    uint32_t skip_crc = 0, skip_crc_want = 0;
    // speculative side
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::map<uint64_t, std::shared_ptr<Member>> members;   // by start offset, all > pos or == pos
    std::deque<std::shared_ptr<Member>> todo;
    std::vector<std::thread> pool;
    size_t buffered = 0;
    // buffers of consumed members, handed to the next worker with their pages still mapped: a 300 MB member in a fresh
    // allocation is 80 000 page faults, and 16 workers faulting at once queue up in the kernel (measured: the parallel
    // reader got SLOWER with the faster decoder until the buffers were recycled)
    std::vector<std::unique_ptr<ByteSink>> spare;
    uint64_t scanned_to = 0;
    bool stop = false;
    // the member currently being served from a worker's buffer
    std::shared_ptr<Member> cur;
    size_t cur_at = 0;
    uint64_t n_par = 0, n_seq = 0;

    void recycle(ByteSink &b)                      // call with mu held
    {
        if (!b.p) return;
        if (spare.size() >= (size_t)threads + 2) { b.release(); return; }
        spare.emplace_back(new ByteSink());
        spare.back()->swap(b);
        spare.back()->n = 0;
    }

    void scan_candidates()                         // call with mu held
    {
        const uint64_t limit = std::min<uint64_t>(size, pos + SCAN_AHEAD);
        uint64_t from = std::max<uint64_t>(scanned_to, pos);         // pos is a member boundary: itself a candidate
        while (from + 10 <= limit) {
            const void *hit = memchr(map + from, 0x1f, limit - 9 - from);
            if (!hit) break;
            const uint64_t at = (const uint8_t *)hit - map;
            const uint8_t *h = map + at;
            // ID1 ID2 CM=8, no reserved flag bits, XFL one of {0, 2, 4}
            if (h[1] == 0x8b && h[2] == 8 && (h[3] & 0xE0) == 0 && (h[8] == 0 || h[8] == 2 || h[8] == 4)) {
                if (!members.count(at)) {
                    auto m = std::make_shared<Member>();
                    m->start = at;
                    members[at] = m;
                    todo.push_back(m);
                }
            }
            from = at + 1;
        }
        scanned_to = std::max<uint64_t>(scanned_to, limit > 9 ? limit - 9 : 0);
        cv_work.notify_all();
    }

    void worker()
    {
        for (;;) {
            std::shared_ptr<Member> m;
            {
                std::unique_lock<std::mutex> lk(mu);
                // (the member the reader is waiting for is never held back by the buffer cap)
                cv_work.wait(lk, [&] { return stop || (!todo.empty() && (buffered < BUFFER_CAP || todo.front()->start <= pos)); });
                if (stop) return;
                m = todo.front();
                todo.pop_front();
                if (m->start < pos) { m->state = 3; continue; }       // the reader has passed it: a false candidate
                m->state = 1;
                if (!spare.empty()) {                                 // the largest spare buffer
                    size_t best = 0;
                    for (size_t i = 1; i < spare.size(); ++i) if (spare[i]->cap > spare[best]->cap) best = i;
                    m->data.swap(*spare[best]);
                    spare.erase(spare.begin() + best);
                }
            }
            const size_t used = inflate_member(map + m->start, size - m->start, m->data, member_cap());   // (only this worker touches m->data)
            {
                std::lock_guard<std::mutex> lk(mu);
                // (a candidate the reader has meanwhile passed was a false one: its bytes must not count as buffered)
                if (used && m->start >= pos) { m->end = m->start + used; m->state = 2; buffered += m->data.n; }
                else { m->state = 3; recycle(m->data); }
            }
            cv_done.notify_all();
        }
    }
};

ParallelGzip::ParallelGzip() : impl_(new Impl()) {}

ParallelGzip::~ParallelGzip()
{
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->stop = true;
    }
    impl_->cv_work.notify_all();
    for (auto &t : impl_->pool) t.join();
    if (impl_->z_open) inflateEnd(&impl_->z);
    if (impl_->map) munmap(const_cast<uint8_t *>(impl_->map), impl_->size);
    if (impl_->fd >= 0) close(impl_->fd);
    delete impl_;
}

std::string ParallelGzip::open(const std::string &path, int threads)
{
    impl_->fd = ::open(path.c_str(), O_RDONLY);
    if (impl_->fd < 0) return "Invalid input file: cannot open " + path;
    struct stat st;
    if (fstat(impl_->fd, &st) != 0) return "Invalid input file: cannot stat " + path;
    impl_->size = (size_t)st.st_size;
    if (impl_->size) {
        void *p = mmap(nullptr, impl_->size, PROT_READ, MAP_PRIVATE, impl_->fd, 0);
        if (p == MAP_FAILED) return "Invalid input file: cannot map " + path;
        impl_->map = static_cast<const uint8_t *>(p);
        madvise(p, impl_->size, MADV_SEQUENTIAL);
    }
    impl_->threads = std::max(1, threads);
    if (impl_->threads > 1) {
        for (int t = 0; t < impl_->threads; ++t) impl_->pool.emplace_back([this] { impl_->worker(); });
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->scan_candidates();
    }
    return "";
}

uint64_t ParallelGzip::members_parallel() const { return impl_->n_par; }
uint64_t ParallelGzip::members_sequential() const { return impl_->n_seq; }

long ParallelGzip::read(uint8_t *dst, size_t n, std::string &err)
{
    Impl &I = *impl_;
    size_t done = 0;
    while (done < n) {
        // 1. bytes of a member a worker has inflated
        if (I.cur) {
            const size_t take = std::min(n - done, I.cur->data.n - I.cur_at);
            if (take) memcpy(dst + done, I.cur->data.p + I.cur_at, take);
            done += take;
            I.cur_at += take;
            if (I.cur_at == I.cur->data.n) {
                std::lock_guard<std::mutex> lk(I.mu);
                I.buffered -= I.cur->data.n;
                I.pos = I.cur->end;
                I.members.erase(I.cur->start);
                I.recycle(I.cur->data);
                // candidates the finished member ran over were false
                while (!I.members.empty() && I.members.begin()->first < I.pos) {
                    auto m = I.members.begin()->second;
                    if (m->state == 2) { I.buffered -= m->data.n; I.recycle(m->data); }
                    I.members.erase(I.members.begin());
                }
                I.cur.reset();
                I.cur_at = 0;
                I.scan_candidates();
            }
            continue;
        }
        // 2. inside a member that is being inflated sequentially
        auto member_ended = [&](uint64_t end) {
            ++I.n_seq;
            if (I.threads > 1) {
                std::lock_guard<std::mutex> lk(I.mu);
                I.pos = end;
                while (!I.members.empty() && I.members.begin()->first < I.pos) {
                    auto m = I.members.begin()->second;
                    if (m->state == 2) { I.buffered -= m->data.n; I.recycle(m->data); }
                    I.members.erase(I.members.begin());
                }
                I.scan_candidates();
            } else I.pos = end;
        };
        if (I.fast_open) {
            const char *fe = nullptr;
            const long r = I.fast.read(I.map, I.size, I.fast_ip, dst + done, n - done, &fe);
            bool ok = r >= 0;
            if (ok) {
                done += (size_t)r;
                I.delivered += (uint64_t)r;
                if (I.test_fail_after >= 0 && I.delivered >= (uint64_t)I.test_fail_after && !I.fast.done()) ok = false;   // (tests: hand over mid-member)
            }
            if (ok) {
                if (I.fast.done()) {
                    const uint8_t *t = I.map + I.fast_ip;
                    ok = I.fast_ip + 8 <= I.size &&
                         ((uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24)) == I.fast.crc() &&
                         ((uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24)) == (uint32_t)I.fast.total();
                    if (ok) { I.fast_open = false; member_ended(I.fast_ip + 8); }
                }
            }
            if (!ok) {                                             // zlib decides: from the member's start, past what went out
                I.fast_open = false;
                memset(&I.z, 0, sizeof I.z);
                if (inflateInit2(&I.z, 15 + 16) != Z_OK) { err = "zlib: inflateInit2 failed"; return -1; }
                I.z_open = true;
                I.in_pos = I.pos;
                I.z_skip = I.delivered;
                // what went out already came from the fast decoder UNVERIFIED (its CRC is only checked at the member's end):
                // zlib's version of those bytes must have the same CRC, or the caller holds wrong bytes — an error, not a fallback
                I.skip_crc_want = I.fast.crc();
                I.skip_crc = 0;
            }
            continue;
        }
        if (I.z_open) {
            if (I.z.avail_in == 0) {
                const size_t chunk = std::min<size_t>(I.size - I.in_pos, 1u << 30);
                if (chunk == 0) { err = "Invalid input file: truncated gzip stream"; return -1; }
                I.z.next_in = const_cast<Bytef *>(I.map + I.in_pos);
                I.z.avail_in = (uInt)chunk;
                I.in_pos += chunk;
            }
            uint8_t scratch[1 << 14];
            const bool skipping = I.z_skip > 0;
            I.z.next_out = skipping ? scratch : dst + done;
            I.z.avail_out = skipping ? (uInt)std::min<uint64_t>(I.z_skip, sizeof scratch) : (uInt)std::min<size_t>(n - done, 1u << 30);
            const uInt before = I.z.avail_out;
            const int rc = inflate(&I.z, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { err = "Invalid input file: corrupt gzip stream"; return -1; }
            if (skipping) {
                I.skip_crc = crc32_fast(I.skip_crc, scratch, before - I.z.avail_out);
                I.z_skip -= before - I.z.avail_out;
                if (I.z_skip == 0 && I.skip_crc != I.skip_crc_want) {
                    err = "Invalid input file: the fast inflate path and zlib disagree on bytes already handed out (decoder defect or memory corruption)";
                    return -1;
                }
            } else done += before - I.z.avail_out;
            if (rc == Z_STREAM_END) {
                if (I.z_skip) { err = "Invalid input file: corrupt gzip stream"; return -1; }   // (shorter than what the fast decoder produced)
                const uint64_t end = I.in_pos - I.z.avail_in;    // the member ended here
                inflateEnd(&I.z);
                I.z_open = false;
                member_ended(end);
            }
            continue;
        }
        // 3. at a member boundary
        if (I.pos >= I.size) break;                                // end of data
        if (I.threads > 1) {
            std::unique_lock<std::mutex> lk(I.mu);
            auto it = I.members.find(I.pos);
            if (it != I.members.end()) {
                auto m = it->second;
                I.cv_done.wait(lk, [&] { return m->state >= 2; });
                if (m->state == 2) { I.cur = m; I.cur_at = 0; ++I.n_par; continue; }
                I.members.erase(it);                               // too big for a worker (or broken): sequentially, from here
            }
        }
        // bytes after a complete member that do not start another one (tar / block padding) end the data, as they do for
        // zlib's gz* readers and for the in-memory path (gunzip_members) — the same with any number of threads
        if (I.n_par + I.n_seq > 0 && (I.size - I.pos < 18 || I.map[I.pos] != 0x1f || I.map[I.pos + 1] != 0x8b)) break;
        I.delivered = 0;
        I.z_skip = 0;
        const size_t hl = I.no_fast ? 0 : gzip_header_length(I.map + I.pos, I.size - I.pos);
        if (hl) {
            I.fast.begin();
            I.fast.test_flip_output_byte(I.test_flip_at);
            I.fast_ip = I.pos + hl;
            I.fast_open = true;
            continue;
        }
        memset(&I.z, 0, sizeof I.z);
        if (inflateInit2(&I.z, 15 + 16) != Z_OK) { err = "zlib: inflateInit2 failed"; return -1; }
        I.z_open = true;
        I.in_pos = I.pos;
    }
    return (long)done;
}

}  // namespace lashhost
