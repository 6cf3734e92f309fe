#include "sketch_files.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "../../../include/lash_gfx950.h"
#include "fastx.hpp"
#include "json_out.hpp"
#include "zstd_dl.hpp"

namespace lashhost {

std::string layout_from_option(const std::string &spec, lash_layout &out)
{
    const char *env = getenv("LASH_LAYOUT");
    const std::string text = !spec.empty() ? spec : (env ? env : "");
    if (lash_layout_parse(text.c_str(), &out) != LASH_OK)
        return "bad layout '" + text + "' (codes=ACGT,kmer=msb|lsb,hmh_x=high|low,hmh_reg=le|be,hll_bucket=low|high,fastq_err=stop|skip,"
               "hmh_hdr=,hll_hdr=azspl,ull_hdr=l)";
    return "";
}

std::string read_list_file(const std::string &path, std::vector<std::string> &files)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) return "cannot open list file " + path;
    std::string line;
    files.clear();
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();       // BufRead::lines strips "\r\n" too
        bool blank = true;
        for (unsigned char c : line)
            if (!(c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r')) { blank = false; break; }
        if (!blank) files.push_back(line);                                // kept verbatim (main.rs:204-206)
    }
    return "";
}

std::string write_parameters_json(const std::string &output_name, const std::string &algorithm, int k, int precision,
                                  uint64_t seed, bool amino)
{
    std::map<std::string, std::string> kv;
    kv["k"] = std::to_string(k);
    kv["algorithm"] = algorithm;
    kv["seed"] = std::to_string(seed);
    kv["molecule"] = amino ? "amino_acid" : "nucleotide";                 // main.rs:248-252 (the reference hard-wires aa = false, main.rs:198)
    if (algorithm == "ull" || algorithm == "hll") kv["precision"] = std::to_string(precision);
    std::ofstream out(output_name + "_parameters.json", std::ios::binary);
    if (!out) return "cannot create " + output_name + "_parameters.json";
    out << json_pretty_string_object(kv);
    return out.good() ? "" : "write failed";
}

namespace {

// page-locked host memory (H2D at PCIe speed); recycled between batches
struct PinnedBuf {
    uint8_t *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t n)
    {
        if (n <= cap) return true;
        lash_host_free_pinned(p);
        cap = n + n / 8 + (1u << 20);
        p = static_cast<uint8_t *>(lash_host_alloc_pinned(cap));
        if (!p) cap = 0;
        return p != nullptr;
    }
    ~PinnedBuf() { lash_host_free_pinned(p); }
};

struct FileSlot {
    bool compressed = false, sized = false, big = false;
    uint64_t size = 0;
    std::vector<uint8_t> inflated;        // compressed inputs only, until placed
    std::string err;
};

struct Batch {
    uint64_t index = 0;
    size_t f0 = 0, f1 = 0;                // files [f0, f1)
    std::unique_ptr<PinnedBuf> buf;
    std::vector<uint64_t> file_off{0};
    std::vector<uint8_t> fmt;
    std::atomic<size_t> remaining{0};
    std::vector<uint8_t> images;
    std::string err;
    std::mutex emu;
};

// simple worker pool
class Pool {
public:
    explicit Pool(int n)
    {
        for (int i = 0; i < n; ++i)
            th_.emplace_back([this] {
                for (;;) {
                    std::function<void()> f;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                        if (q_.empty()) return;
                        f = std::move(q_.front());
                        q_.pop_front();
                    }
                    f();
                }
            });
    }
    void submit(std::function<void()> f)
    {
        std::lock_guard<std::mutex> lk(mu_);
        q_.push_back(std::move(f));
        cv_.notify_one();
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
private:
    std::vector<std::thread> th_;
    std::deque<std::function<void()>> q_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};

bool peek_compressed(const std::string &path, uint64_t &size, std::string &err)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) { err = "Invalid input file: cannot open " + path; return false; }
    size = (uint64_t)st.st_size;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "Invalid input file: cannot open " + path; return false; }
    unsigned char m[6] = {0, 0, 0, 0, 0, 0};
    size_t got = fread(m, 1, 6, f);
    fclose(f);
    if (got >= 2 && m[0] == 0x1f && m[1] == 0x8b) return true;
    if (got >= 4 && m[0] == 0x28 && m[1] == 0xb5 && m[2] == 0x2f && m[3] == 0xfd) return true;
    if (got >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') return true;
    if (got >= 6 && m[0] == 0xfd && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z' && m[5] == 0) return true;
    return false;
}

std::string read_exact(const std::string &path, uint8_t *dst, uint64_t size)
{
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return "Invalid input file: cannot open " + path;
    uint64_t done = 0;
    while (done < size) {
        ssize_t n = pread(fd, dst + done, (size_t)std::min<uint64_t>(size - done, 1u << 30), (off_t)done);
        if (n <= 0) { close(fd); return "Invalid input file: short read on " + path; }
        done += (uint64_t)n;
    }
    close(fd);
    return "";
}

// needletail sniffs '>' / '@' (SURVEY App. A.5); anything else is "Invalid input file" (utils.rs:453)
// needletail decides by the FIRST byte of the (decompressed) file and fails on anything else (parse_fastx_file, utils.rs:453)
int sniff_format(const uint8_t *p, uint64_t n)
{
    if (n == 0) return 0;
    return p[0] == '>' ? LASH_FMT_FASTA : p[0] == '@' ? LASH_FMT_FASTQ : 0;
}

// Where to cut a chunk of a large file so that the next chunk starts at a record boundary.
//   FASTA: before the last "\n>" in the second half; a record longer than the chunk is cut at a line boundary and
//          `overlap` bytes (whole lines, >= 4 KiB, header-free by construction) are repeated at the start of the next
//          chunk so that no k-mer across the cut is lost (repeated k-mers are harmless: max / OR are idempotent).
//   FASTQ: before the last line that starts with '@' and whose line-after-next starts with '+' (a quality line that
//          happens to start with '@' fails that test).
// Where to end this chunk of a file that is streamed in pieces, and what the next chunk has to start with.
// FASTA: before the last '\n>' of the second half when there is one (nothing to carry).  Otherwise the record continues
// in the next chunk: cut at the last line end (or at the chunk end inside one enormous line) and carry the last <= 32
// SURVIVING bases of the current record as a synthetic sequence line.  Those are exactly the bases a k-mer spanning the
// cut can reach back to (k - 1 <= 31; filter_out_n deletes everything else anyway), so the k-mer set of the pieces
// equals that of the whole record however long an N run at the cut is; the few k-mers inside the carried line are
// repeats, which max / OR ignore.  FASTQ: before the last complete record start of the second half; nothing carried.
size_t find_cut(const uint8_t *b, size_t n, int fmt, std::vector<uint8_t> &carry)
{
    carry.clear();
    if (fmt == LASH_FMT_FASTA) {
        for (size_t q = n - 1; q > n / 2; --q)
            if (b[q] == '>' && b[q - 1] == '\n') return q;
        size_t cut = n;
        while (cut > n / 2 && b[cut - 1] != '\n') --cut;
        if (cut <= n / 2) cut = n;                        // one enormous line: cut inside it
        // walk back line by line from the cut, collecting surviving bases, until 32 are found or a header line is met
        std::vector<uint8_t> rev;                         // collected bases, last one first
        size_t line_stop = cut;                           // one past the last byte of the line being looked at
        while (rev.size() < 32 && line_stop > 0) {
            size_t ls = line_stop;                        // start of that line
            if (ls > 0 && b[ls - 1] == '\n') --ls;        // (step over the terminator of the previous line)
            while (ls > 0 && b[ls - 1] != '\n') --ls;
            if (b[ls] == '>') break;                      // the record began here: nothing before it belongs to it
            for (size_t i = line_stop; i > ls && rev.size() < 32; --i) {
                const uint8_t ch = b[i - 1];
                if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') rev.push_back(ch);
            }
            line_stop = ls;
        }
        if (!rev.empty()) {
            carry.assign(rev.rbegin(), rev.rend());
            carry.push_back('\n');
        }
        return cut;
    }
    auto line_end = [&](size_t p) { const void *e = memchr(b + p, '\n', n - p); return e ? (size_t)((const uint8_t *)e - b) : n; };
    for (size_t q = n - 1; q > n / 2; --q) {
        if (b[q] != '@' || b[q - 1] != '\n') continue;
        const size_t e1 = line_end(q);
        if (e1 >= n) continue;
        const size_t e2 = line_end(e1 + 1);
        if (e2 >= n || e2 + 1 >= n) continue;
        if (b[e2 + 1] == '+') return q;
    }
    return 0;                                             // no FASTQ record boundary in the second half
}

// Two pinned chunk buffers: this thread fills chunk n+1 (inflate wait + copy out of the members + finding the cut)
// while a second thread has the library sketch chunk n (H2D copy + kernels; the only user of `ctx` meanwhile).
std::string stream_big_file(lash_ctx *ctx, const lash_params &prm0, const std::string &path, uint64_t chunk_bytes,
                            PinnedBuf &buf0, PinnedBuf &buf1, uint8_t *image, uint64_t &bytes_seen, int threads)
{
    ByteStream bs;
    bs.set_threads(threads);                              // multi-member .gz: members inflate in parallel (pgzip.hpp)
    std::string err = bs.open(path);
    if (!err.empty()) return err;
    if (!buf0.reserve(chunk_bytes + 64) || !buf1.reserve(chunk_bytes + 64)) return "out of pinned host memory";
    PinnedBuf *bufs[2] = {&buf0, &buf1};

    // ---- the sketching side ----
    struct Job { int b; size_t cut; int fmt; bool force; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    bool busy[2] = {false, false}, no_more = false, halt = false, corner_noted = false;
    // HyperLogLog: the image before each chunk and the carried incremental `sum` (lash_hll_replay_streamed_chunk, round 5: a streamed
    // file's header then equals the reference's, which reads the whole file into one sketch in order, utils.rs:457-505)
    std::vector<uint8_t> image_before;
    double hll_carry[2] = {0.0, 0.0};
    int hll_have_carry = 0;
    lash_layout lay0;
    (void)lash_ctx_get_layout(ctx, &lay0);
    const bool skip_bad = lay0.fastq_skip_bad != 0;           // layout switch U6: malformed records are dropped, the reading goes on
    std::string gpu_err;
    uint64_t calls = 0;
    double t_gpu = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(now() - t0).count(); };
    std::thread sketcher([&]() {
        for (;;) {
            Job jb;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !jobs.empty() || no_more; });
                if (jobs.empty()) return;
                jb = jobs.front();
                jobs.pop_front();
            }
            bool stop = false;
            std::string e;
            {
                bool skip;
                { std::lock_guard<std::mutex> lk(mu); skip = halt; }
                if (!skip && (jb.cut || jb.force)) {          // (a file whose FIRST record is malformed still owes its empty sketch)
                    const auto t0 = now();
                    lash_params prm = prm0;
                    if (calls) prm.flags |= LASH_F_ACCUMULATE;
                    const uint64_t off[2] = {0, (uint64_t)jb.cut};
                    const uint8_t f = (uint8_t)jb.fmt;
                    const bool hll_nt = prm.algo == LASH_HLL && !(prm.flags & LASH_F_AMINO);
                    const size_t ib = hll_nt ? lash_layout_image_bytes(&lay0, LASH_HLL, prm.p) : 0;
                    if (hll_nt) {
                        if (calls) image_before.assign(image, image + ib);
                        else {                                    // the first chunk starts from the empty sketch: every register 0
                            image_before.assign(ib, 0);
                        }
                    }
                    const int rc = lash_sketch_files_raw(ctx, &prm, bufs[jb.b]->p, off, &f, 1, image);
                    if (rc != LASH_OK) e = std::string(lash_strerror(rc)) + " " + lash_ctx_last_error(ctx);
                    else {
                        ++calls;
                        if (hll_nt) {
                            // a register above 53 - p: this chunk's part of the incremental sum is replayed, later chunks carry it on
                            const int rr = lash_hll_replay_streamed_chunk(ctx, &prm0, bufs[jb.b]->p, (uint64_t)jb.cut, jb.fmt, image_before.data(), image,
                                                                          hll_carry, &hll_have_carry);
                            if (rr != LASH_OK && !corner_noted) {
                                corner_noted = true;
                                fprintf(stderr, "note: %s: a HyperLogLog register exceeds 53 - p and the replay of the streamed chunk failed (%s); the header's "
                                                "sum field is the exact sum and may differ from lash's incrementally rounded value in its last bits\n",
                                        path.c_str(), lash_strerror(rr));
                            }
                        }
                        // a malformed FASTQ record ends needletail's iteration (utils.rs:457): the library kept this chunk's
                        // records before it; nothing after it belongs to the sketch
                        if (lash_ctx_format_errors(ctx, nullptr, 0) != 0 && !skip_bad) stop = true;
                    }
                    t_gpu += since(t0);
                }
            }
            std::lock_guard<std::mutex> lk(mu);
            if (!e.empty() && gpu_err.empty()) gpu_err = e;
            if (stop || !e.empty()) halt = true;
            busy[jb.b] = false;
            cv.notify_all();
        }
    });
    auto finish = [&](std::string result) {
        { std::lock_guard<std::mutex> lk(mu); no_more = true; }
        cv.notify_all();
        sketcher.join();
        if (result.empty() && !gpu_err.empty()) result = gpu_err;
        return result;
    };

    // ---- the reading side ----
    int cur = 0;
    size_t have = 0;                                      // bytes at the front of the current buffer carried from the previous chunk
    bool first = true, eof = false;
    int fmt = 0;
    // LASH_CLI_TIMING: where a streamed file's wall time goes (read = inflate wait + copy out of the members; check; waiting
    // for the sketching side; its own time runs concurrently)
    const bool timing = getenv("LASH_CLI_TIMING") != nullptr;
    double t_read = 0, t_check = 0, t_wait = 0;
    std::string result;
    while (!eof) {
        uint8_t *b = bufs[cur]->p;
        auto t0 = now();
        while (have < chunk_bytes) {
            const long r = bs.read(b + have, chunk_bytes - have, err);
            if (r < 0) { result = err + " (" + path + ")"; break; }
            if (r == 0) { eof = true; break; }
            have += (size_t)r;
            bytes_seen += (uint64_t)r;
        }
        if (!result.empty()) break;
        t_read += since(t0);
        if (first) {
            fmt = sniff_format(b, have);
            if (!fmt) { result = "Invalid input file: neither FASTA ('>') nor FASTQ ('@'): " + path; break; }
        }
        t0 = now();
        std::vector<uint8_t> carry;
        size_t cut = eof ? have : find_cut(b, have, fmt, carry);
        if ((prm0.flags & LASH_F_AMINO) && !carry.empty()) { result = "a protein record larger than a --stream-mb chunk: " + path; break; }
        if (!eof && cut == 0) { result = "cannot find a record boundary inside a " + std::to_string(chunk_bytes >> 20) + " MiB chunk of " + path; break; }
        // FASTQ: the chunk starts and ends at record boundaries and goes to the library as it is.  Malformed records — line
        // structure AND quality-line lengths — are found on the device (pack_kernels.hip, fastq_check.hip); the library then
        // re-does the chunk with needletail's rule and reports it (round 3: no host pass over every byte any more).
        const bool stop_here = false;
        t_check += since(t0);
        // what follows the cut moves to the front of the other buffer — once the sketching side has let go of it
        t0 = now();
        const int other = cur ^ 1;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !busy[other]; });
            if (halt) break;                              // an error, or the library found the record structure broken: nothing more to add
        }
        t_wait += since(t0);
        const size_t rest = have - cut;                   // cut > have / 2, carry <= 33 bytes: the buffer always drains
        if (!stop_here && !eof) {
            if (!carry.empty()) memcpy(bufs[other]->p, carry.data(), carry.size());
            if (rest) memcpy(bufs[other]->p + carry.size(), b + cut, rest);
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            busy[cur] = true;
            jobs.push_back(Job{cur, cut, fmt, stop_here && first});
        }
        cv.notify_all();
        first = false;
        if (stop_here) break;
        have = carry.size() + rest;
        cur = other;
    }
    result = finish(result);
    if (timing)
        fprintf(stderr, "[lash cli] streamed %s: read %.2f s, check %.2f s, waited for the sketching side %.2f s (its calls: %.2f s, concurrent)\n",
                path.c_str(), t_read, t_check, t_wait, t_gpu);
    if (result.empty() && calls == 0) return "Invalid input file: empty (" + path + ")";
    return result;
}

}  // namespace

// test hook (host_hooks.cpp): the chunk-cut rule of the large-file streamer
size_t stream_find_cut(const uint8_t *b, size_t n, int fmt, std::vector<uint8_t> &carry) { return find_cut(b, n, fmt, carry); }

std::string sketch_files(const SketchOptions &opt, const std::vector<std::string> &files, const std::string &output_name,
                         SketchStats *stats)
{
    const auto t_start = std::chrono::steady_clock::now();
    // LASH_CLI_TIMING=1: pipeline marks on stderr (seconds since sketch_files() started)
    const bool timing = getenv("LASH_CLI_TIMING") != nullptr;
    std::mutex tmu;
    auto mark = [&](const char *what, uint64_t a = 0, double dur = -1.0) {
        if (!timing) return;
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        std::lock_guard<std::mutex> lk(tmu);
        if (dur >= 0) fprintf(stderr, "[lash cli] %7.3f s  %s %llu (%.3f s)\n", t, what, (unsigned long long)a, dur);
        else fprintf(stderr, "[lash cli] %7.3f s  %s %llu\n", t, what, (unsigned long long)a);
    };
    lash_params prm{opt.algo, opt.k, opt.precision, opt.flags, opt.seed};
    if (lash_params_check(&prm) != LASH_OK) return lash_strerror(LASH_EINVAL);
    const size_t ib = lash_layout_image_bytes(&opt.layout, opt.algo, opt.precision);
    std::vector<int> devices = opt.devices.empty() ? std::vector<int>{0} : opt.devices;
    const int n_dev_avail = lash_device_count();
    if (n_dev_avail <= 0) return lash_strerror(LASH_ENODEV);
    for (int d : devices)
        if (d < 0 || d >= n_dev_avail) return "device index out of range";
    const size_t n_files = files.size();

    // ---- GPU workers: one context per device; the FASTA/FASTQ parse runs on the GPU (lash_sketch_files_raw) ----
    std::deque<std::shared_ptr<Batch>> todo;
    std::map<uint64_t, std::shared_ptr<Batch>> finished;
    std::vector<std::unique_ptr<PinnedBuf>> pool_bufs;      // recycled pinned buffers (guarded by qmu)
    std::mutex qmu;
    std::condition_variable cv_todo, cv_done;
    bool no_more = false;
    size_t in_flight = 0;                                  // batches planned but not yet written (guarded by qmu)
    // Enough batches to keep every device fed and to read ahead while the HIP contexts come up (~0.3 s), but not more
    // page-locked memory than ~512 MiB: pinning costs ~0.18 s per GB and is the slowest thing the host does here.
    const size_t in_flight_cap = std::max<size_t>(2 * devices.size() + 1,
                                                  (size_t)((512ull << 20) / std::max<uint64_t>(opt.batch_bytes, 1)));
    auto gpu_worker = [&](int device) {
        lash_ctx *ctx = nullptr;
        int rc = lash_ctx_create(&ctx, device);
        if (rc == LASH_OK) rc = lash_ctx_set_layout(ctx, &opt.layout);
        mark("context ready on device", (uint64_t)device);
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> lk(qmu);
                cv_todo.wait(lk, [&] { return !todo.empty() || no_more; });
                if (todo.empty()) break;
                b = todo.front();
                todo.pop_front();
            }
            if (b->err.empty()) {
                if (rc != LASH_OK) b->err = lash_strerror(rc);
                else {
                    const uint32_t ng = (uint32_t)(b->f1 - b->f0);
                    b->images.assign((size_t)ng * ib, 0);
                    const auto g0 = std::chrono::steady_clock::now();
                    int r2 = lash_sketch_files_raw(ctx, &prm, b->buf->p, b->file_off.data(), b->fmt.data(), ng, b->images.data());
                    mark("GPU done, batch", b->index, std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count());
                    if (r2 != LASH_OK) b->err = std::string(lash_strerror(r2)) + " " + lash_ctx_last_error(ctx);
                    else if (prm.algo == LASH_HLL) {
                        // a register above 53 - p: the image's `sum` field is the exact sum, the reference's incrementally
                        // rounded one may differ in its last bits (include/lash_gfx950.h: lash_ctx_hll_inexact_sums)
                        std::vector<uint32_t> idx(ng);
                        const uint32_t nc = lash_ctx_hll_inexact_sums(ctx, idx.data(), ng);
                        for (uint32_t i = 0; i < nc && i < ng; ++i)
                            fprintf(stderr, "note: %s: a HyperLogLog register exceeds 53 - p; the header's sum field is the exact sum and may "
                                            "differ from lash's incrementally rounded value in its last bits\n", files[b->f0 + idx[i]].c_str());
                    }
                }
            }
            std::lock_guard<std::mutex> lk(qmu);
            if (b->buf) pool_bufs.push_back(std::move(b->buf));
            finished[b->index] = b;
            cv_done.notify_all();
        }
        if (ctx) lash_ctx_destroy(ctx);
    };
    std::vector<std::thread> workers;
    for (int d : devices) workers.emplace_back(gpu_worker, d);

    // ---- writer: images in batch (== file) order into one zstd stream (utils.rs:567-574; frames: zstd_dl.hpp) ----
    std::string werr;
    uint64_t n_batches_total = 0;
    bool batches_known = false;
    std::thread writer([&]() {
        ZstdWriter zw;
        werr = zw.open(output_name + "_sketches.bin", 3, opt.threads);     // Encoder::new(w, 3) + multithread(threads)
        uint64_t want = 0;
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> lk(qmu);
                cv_done.wait(lk, [&] { return finished.count(want) || (batches_known && want >= n_batches_total); });
                if (!finished.count(want)) break;
                b = finished[want];
                finished.erase(want);
                --in_flight;
                cv_done.notify_all();
            }
            if (werr.empty() && !b->err.empty()) werr = b->err;
            if (werr.empty()) werr = zw.write(b->images.data(), b->images.size());
            ++want;
        }
        if (werr.empty()) werr = zw.finish();
    });

    // ---- readers: inflate compressed inputs ahead of the planner; place every file's bytes into its batch ----
    std::vector<FileSlot> slots(n_files);
    std::mutex smu;
    std::condition_variable cv_sized, cv_window;
    uint64_t inflated_held = 0;                            // bytes of inflated-but-unplaced data (guarded by smu)
    const uint64_t inflate_window = std::max<uint64_t>(2 * opt.batch_bytes, 1ull << 28);
    std::string err;
    uint64_t n_bytes = 0, batch_index = 0;
    {
        Pool pool(std::max(1, opt.threads));
        size_t next_inflate = 0;
        auto pump_inflates = [&]() {                       // called with smu held
            while (next_inflate < n_files) {
                FileSlot &s = slots[next_inflate];
                if (!s.compressed) { ++next_inflate; continue; }
                if (inflated_held >= inflate_window) break;
                const size_t i = next_inflate++;
                inflated_held += 1;                        // placeholder so that at least progress is bounded per file
                pool.submit([&, i]() {
                    std::vector<uint8_t> data;
                    std::string e = slurp_maybe_compressed(files[i], data);
                    std::lock_guard<std::mutex> lk(smu);
                    slots[i].err = e;
                    slots[i].size = data.size();
                    inflated_held += data.size();
                    slots[i].inflated.swap(data);
                    slots[i].sized = true;
                    cv_sized.notify_all();
                });
            }
        };
        const uint64_t stream_bytes = std::min<uint64_t>(std::max<uint64_t>(opt.stream_bytes, 1u << 16), 0xF0000000ull);
        {
            // size and kind of every file, on the pool's threads in ranges of 512 (a collection of 100 000 viral genomes spent 0.25 s
            // in this loop on one thread — as long as the GPU needs for all of them a hundred times over); the first error in FILE
            // order is the one reported, as the sequential loop did
            const size_t R = 512, n_ranges = (n_files + R - 1) / R;
            std::vector<std::string> range_err(n_ranges);
            std::mutex pmu;
            std::condition_variable pcv;
            size_t pending = n_ranges;
            for (size_t r = 0; r < n_ranges; ++r) {
                pool.submit([&, r]() {
                    for (size_t i = r * R; i < std::min(n_files, (r + 1) * R); ++i) {
                        std::string e;
                        uint64_t sz = 0;
                        const bool comp = peek_compressed(files[i], sz, e);
                        if (!e.empty()) { range_err[r] = e; break; }
                        slots[i].compressed = comp;
                        slots[i].big = comp ? sz > stream_bytes / 3 : sz > stream_bytes;
                        if (slots[i].big) { slots[i].compressed = false; slots[i].size = 0; slots[i].sized = true; }   // handled by the planner itself
                        else if (!comp) { slots[i].size = sz; slots[i].sized = true; }
                    }
                    std::lock_guard<std::mutex> lk(pmu);
                    if (--pending == 0) pcv.notify_all();
                });
            }
            std::unique_lock<std::mutex> lk(pmu);
            pcv.wait(lk, [&] { return pending == 0; });
            for (size_t r = 0; r < n_ranges && err.empty(); ++r) err = range_err[r];
        }
        lash_ctx *stream_ctx = nullptr;
        PinnedBuf stream_buf, stream_buf2;          // the streamer's two chunk buffers, kept across files (pinning costs 0.2 s per GiB)
        std::shared_ptr<Batch> cur;
        auto finalize = [&]() {
            if (!cur || cur->f1 == cur->f0) return;
            {   // bounded number of batches planned / queued / running / unwritten
                std::unique_lock<std::mutex> lk(qmu);
                cv_done.wait(lk, [&] { return in_flight < in_flight_cap; });
                ++in_flight;
                if (!pool_bufs.empty()) { cur->buf = std::move(pool_bufs.back()); pool_bufs.pop_back(); }
            }
            if (!cur->buf) cur->buf.reset(new PinnedBuf());
            const auto p0 = std::chrono::steady_clock::now();
            if (!cur->buf->reserve(cur->file_off.back() + 64)) { cur->err = "out of pinned host memory"; }
            cur->index = batch_index++;
            mark("batch planned (pinned buffer ready)", cur->index, std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count());
            cur->fmt.assign(cur->f1 - cur->f0, 0);
            cur->remaining = cur->f1 - cur->f0;
            std::shared_ptr<Batch> b = cur;
            // one task per run of files of about 1 MiB (at most 256 files): a task per 10 kB file is 2 us of work behind a queue that
            // sixteen threads share — 6 250 files of a batch took 32 ms to read, 80 us per file and thread
            for (size_t g0 = b->f0; g0 < b->f1;) {
                size_t g1 = g0 + 1;
                uint64_t run = slots[g0].size;
                while (g1 < b->f1 && g1 - g0 < 256 && run + slots[g1].size <= (1u << 20)) run += slots[g1++].size;
                pool.submit([&, b, g0, g1]() {
                  for (size_t i = g0; i < g1; ++i) {
                    FileSlot &s = slots[i];
                    uint8_t *dst = b->buf->p ? b->buf->p + b->file_off[i - b->f0] : nullptr;
                    std::string e = s.err;
                    if (e.empty() && dst) {
                        if (s.compressed) {
                            if (s.size) memcpy(dst, s.inflated.data(), s.size);
                            std::vector<uint8_t>().swap(s.inflated);
                            std::lock_guard<std::mutex> lk(smu);
                            inflated_held -= std::min<uint64_t>(inflated_held, s.size + 1);
                            cv_window.notify_all();
                        } else {
                            e = read_exact(files[i], dst, s.size);
                        }
                        if (e.empty()) {
                            const int f = sniff_format(dst, s.size);
                            if (!f) e = "Invalid input file: neither FASTA ('>') nor FASTQ ('@'): " + files[i];
                            b->fmt[i - b->f0] = (uint8_t)(f ? f : LASH_FMT_FASTA);
                            // (malformed FASTQ records: found on the device, the file is then re-done by the library with
                            // needletail's rule — utils.rs:457 — or, under layout fastq_err=skip, without the bad records)
                        }
                    }
                    if (!e.empty()) { std::lock_guard<std::mutex> lk(b->emu); if (b->err.empty()) b->err = e; }
                    if (b->remaining.fetch_sub(1) == 1) {
                        mark("batch read into memory", b->index);
                        std::lock_guard<std::mutex> lk(qmu);
                        todo.push_back(b);
                        cv_todo.notify_one();
                    }
                  }
                });
                g0 = g1;
            }
            cur.reset();
        };
        for (size_t i = 0; i < n_files && err.empty(); ++i) {
            {
                std::unique_lock<std::mutex> lk(smu);
                pump_inflates();
                while (!slots[i].sized) {
                    cv_sized.wait_for(lk, std::chrono::milliseconds(50));
                    pump_inflates();
                }
            }
            if (slots[i].big) {
                // a large file is its own "batch": streamed here, in file order, into one accumulated image
                finalize();
                auto b = std::make_shared<Batch>();
                b->f0 = i;
                b->f1 = i + 1;
                b->images.assign(ib, 0);
                if (!stream_ctx) {
                    int rc = lash_ctx_create(&stream_ctx, devices[0]);
                    if (rc == LASH_OK) rc = lash_ctx_set_layout(stream_ctx, &opt.layout);
                    if (rc != LASH_OK) { err = lash_strerror(rc); break; }
                }
                uint64_t seen = 0;
                b->err = stream_big_file(stream_ctx, prm, files[i], stream_bytes, stream_buf, stream_buf2, b->images.data(), seen, opt.threads);
                n_bytes += seen;
                std::lock_guard<std::mutex> lk(qmu);
                b->index = batch_index++;
                ++in_flight;
                finished[b->index] = b;
                cv_done.notify_all();
                continue;
            }
            if (!cur) { cur = std::make_shared<Batch>(); cur->f0 = cur->f1 = i; }
            cur->file_off.push_back(cur->file_off.back() + slots[i].size);
            cur->f1 = i + 1;
            n_bytes += slots[i].size;
            if (cur->file_off.back() >= opt.batch_bytes) finalize();
        }
        if (err.empty()) finalize();
        if (stream_ctx) lash_ctx_destroy(stream_ctx);
    }   // Pool joins here: every placement task has run
    {
        std::lock_guard<std::mutex> lk(qmu);
        no_more = true;
        n_batches_total = batch_index;
        batches_known = true;
        cv_todo.notify_all();
        cv_done.notify_all();
    }
    for (auto &t : workers) t.join();
    {
        std::lock_guard<std::mutex> lk(qmu);
        cv_done.notify_all();
    }
    writer.join();
    mark("writer done", 0);
    if (err.empty()) err = werr;
    if (!err.empty()) return err;

    // ---- names (utils.rs:577-580) ----
    {
        std::ofstream out(output_name + "_files.json", std::ios::binary);
        if (!out) return "cannot create " + output_name + "_files.json";
        out << json_pretty_string_array(files);
        if (!out.good()) return "write failed";
    }
    if (stats) {
        stats->files = n_files;
        stats->records = 0;
        stats->bytes = n_bytes;
        stats->batches = batch_index;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    }
    return "";
}

}  // namespace lashhost
