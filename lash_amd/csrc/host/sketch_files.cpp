#include "sketch_files.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "../../../include/lash_gfx950.h"
#include "fastx.hpp"
#include "json_out.hpp"
#include "zstd_dl.hpp"

namespace lashhost {

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
                                  uint64_t seed)
{
    std::map<std::string, std::string> kv;
    kv["k"] = std::to_string(k);
    kv["algorithm"] = algorithm;
    kv["seed"] = std::to_string(seed);
    kv["molecule"] = "nucleotide";                                        // aa is hard-wired false (main.rs:198)
    if (algorithm == "ull" || algorithm == "hll") kv["precision"] = std::to_string(precision);
    std::ofstream out(output_name + "_parameters.json", std::ios::binary);
    if (!out) return "cannot create " + output_name + "_parameters.json";
    out << json_pretty_string_object(kv);
    return out.good() ? "" : "write failed";
}

namespace {

struct ParsedFile {
    RecordBatch rb;
    std::string err;
    bool done = false;
};

// sequence bytes of one batch in page-locked memory (H2D at PCIe speed); buffers are recycled between batches
struct PinnedBuf {
    uint8_t *p = nullptr;
    size_t cap = 0, size = 0;
    bool append(const uint8_t *src, size_t n)
    {
        if (size + n > cap) {
            size_t want = std::max(size + n, cap + cap / 2 + (64u << 20));
            uint8_t *q = static_cast<uint8_t *>(lash_host_alloc_pinned(want));
            if (!q) return false;
            if (size) memcpy(q, p, size);
            lash_host_free_pinned(p);
            p = q;
            cap = want;
        }
        if (n) memcpy(p + size, src, n);
        size += n;
        return true;
    }
    ~PinnedBuf() { lash_host_free_pinned(p); }
};

struct Batch {
    uint64_t index = 0;
    std::unique_ptr<PinnedBuf> seq;
    std::vector<uint64_t> rec_off{0};
    std::vector<uint64_t> genome_rec_off{0};
    std::vector<uint8_t> images;
    std::string err;
};

}  // namespace

std::string sketch_files(const SketchOptions &opt, const std::vector<std::string> &files, const std::string &output_name,
                         SketchStats *stats)
{
    const auto t_start = std::chrono::steady_clock::now();
    lash_params prm{opt.algo, opt.k, opt.precision, opt.flags, opt.seed};
    if (lash_params_check(&prm) != LASH_OK) return lash_strerror(LASH_EINVAL);
    const size_t ib = lash_sketch_image_bytes(opt.algo, opt.precision);
    std::vector<int> devices = opt.devices.empty() ? std::vector<int>{0} : opt.devices;
    const int n_dev_avail = lash_device_count();
    if (n_dev_avail <= 0) return lash_strerror(LASH_ENODEV);
    for (int d : devices)
        if (d < 0 || d >= n_dev_avail) return "device index out of range";
    const int n_readers = std::max(1, opt.threads);
    const size_t n_files = files.size();

    // ---- stage 1: reader pool, files claimed in order, results delivered in order ----
    std::vector<std::unique_ptr<ParsedFile>> parsed(n_files);
    for (auto &p : parsed) p.reset(new ParsedFile());
    std::mutex mu;
    std::condition_variable cv_parsed, cv_window;
    std::atomic<size_t> next_file{0};
    size_t consumed = 0;                                   // files already merged into batches (guarded by mu)
    uint64_t inflight_bytes = 0;                           // parsed but not yet consumed
    const uint64_t window_bytes = std::max<uint64_t>(opt.batch_bytes * 2, 1ull << 28);
    bool abort_all = false;
    auto reader = [&]() {
        for (;;) {
            size_t i = next_file.fetch_add(1);
            if (i >= n_files) return;
            {   // bounded look-ahead so that host memory stays ~2 batches
                std::unique_lock<std::mutex> lk(mu);
                cv_window.wait(lk, [&] { return abort_all || inflight_bytes < window_bytes || i == consumed; });
                if (abort_all) return;
            }
            ParsedFile *pf = parsed[i].get();
            pf->err = read_fastx_file(files[i], pf->rb);
            std::lock_guard<std::mutex> lk(mu);
            inflight_bytes += pf->rb.seq.size();
            pf->done = true;
            cv_parsed.notify_all();
        }
    };
    std::vector<std::thread> readers;
    for (int r = 0; r < n_readers; ++r) readers.emplace_back(reader);

    // ---- stage 3: GPU workers ----
    std::vector<std::unique_ptr<PinnedBuf>> pool;            // recycled pinned buffers (guarded by qmu)
    std::deque<std::shared_ptr<Batch>> todo;
    std::map<uint64_t, std::shared_ptr<Batch>> finished;
    std::mutex qmu;
    std::condition_variable cv_todo, cv_done;
    bool no_more = false;
    auto gpu_worker = [&](int device) {
        lash_ctx *ctx = nullptr;
        int rc = lash_ctx_create(&ctx, device);
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> lk(qmu);
                cv_todo.wait(lk, [&] { return !todo.empty() || no_more; });
                if (todo.empty()) break;
                b = todo.front();
                todo.pop_front();
            }
            if (rc != LASH_OK) b->err = lash_strerror(rc);
            else {
                const uint32_t ng = (uint32_t)(b->genome_rec_off.size() - 1);
                b->images.assign((size_t)ng * ib, 0);
                int r2 = lash_sketch_batch(ctx, &prm, b->seq->p, b->rec_off.data(), b->rec_off.size() - 1,
                                           b->genome_rec_off.data(), ng, b->images.data());
                if (r2 != LASH_OK) b->err = std::string(lash_strerror(r2)) + " " + lash_ctx_last_error(ctx);
            }
            std::lock_guard<std::mutex> lk(qmu);
            b->seq->size = 0;
            pool.push_back(std::move(b->seq));
            finished[b->index] = b;
            cv_done.notify_all();
        }
        if (ctx) lash_ctx_destroy(ctx);
    };
    std::vector<std::thread> workers;
    for (int d : devices) workers.emplace_back(gpu_worker, d);

    // ---- stage 4: writer (in order) ----
    std::string werr;
    uint64_t n_batches_total = 0;                          // set when known (guarded by qmu)
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
                if (batches_known && want >= n_batches_total && !finished.count(want)) break;
                b = finished[want];
                finished.erase(want);
            }
            if (werr.empty() && !b->err.empty()) werr = b->err;
            if (werr.empty()) werr = zw.write(b->images.data(), b->images.size());
            ++want;
        }
        if (werr.empty()) werr = zw.finish();
    });

    // ---- stage 2 (this thread): merge parsed files into batches in file order ----
    std::string err;
    uint64_t n_records = 0, n_bytes = 0, batch_index = 0;
    auto new_batch = [&]() {
        auto b = std::make_shared<Batch>();
        std::lock_guard<std::mutex> lk(qmu);
        if (!pool.empty()) { b->seq = std::move(pool.back()); pool.pop_back(); }
        else b->seq.reset(new PinnedBuf());
        return b;
    };
    auto cur = new_batch();
    auto submit = [&]() {
        cur->index = batch_index++;
        {
            std::lock_guard<std::mutex> lk(qmu);
            todo.push_back(cur);
            cv_todo.notify_one();
        }
        {   // keep at most 2 batches per device queued or running
            std::unique_lock<std::mutex> lk(qmu);
            cv_done.wait(lk, [&] { return todo.size() < 2 * devices.size(); });
        }
        cur = new_batch();
    };
    for (size_t i = 0; i < n_files && err.empty(); ++i) {
        ParsedFile *pf = parsed[i].get();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_parsed.wait(lk, [&] { return pf->done; });
        }
        if (!pf->err.empty()) { err = pf->err; break; }
        const uint64_t base = cur->seq->size;
        if (!cur->seq->append(pf->rb.seq.data(), pf->rb.seq.size())) { err = "out of pinned host memory"; break; }
        for (size_t r = 1; r < pf->rb.rec_off.size(); ++r) cur->rec_off.push_back(base + pf->rb.rec_off[r]);
        cur->genome_rec_off.push_back(cur->rec_off.size() - 1);
        n_records += pf->rb.n_rec();
        n_bytes += pf->rb.seq.size();
        {
            std::lock_guard<std::mutex> lk(mu);
            inflight_bytes -= pf->rb.seq.size();
            consumed = i + 1;
            cv_window.notify_all();
        }
        parsed[i].reset();
        if (cur->seq->size >= opt.batch_bytes) submit();
    }
    if (err.empty() && cur->genome_rec_off.size() > 1) submit();
    {
        std::lock_guard<std::mutex> lk(mu);
        abort_all = !err.empty();
        cv_window.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(qmu);
        no_more = true;
        n_batches_total = batch_index;
        batches_known = true;
        cv_todo.notify_all();
        cv_done.notify_all();
    }
    for (auto &t : readers) t.join();
    for (auto &t : workers) t.join();
    {
        std::lock_guard<std::mutex> lk(qmu);
        cv_done.notify_all();
    }
    writer.join();
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
        stats->records = n_records;
        stats->bytes = n_bytes;
        stats->batches = batch_index;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    }
    return "";
}

}  // namespace lashhost
