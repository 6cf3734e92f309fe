// dist.cpp — `lash dist` for the gfx950 build.
//
// Mirrors /root/reference/src/main.rs:280-617 (file discovery, parameter checks, output formats, distance formula)
// and utils.rs:84-373 (hmh_distance, ull_distance, hll_distance).  The O(N_ref * N_qry * registers) scans run on the GPU(s);
// what is O(sketches) or O(pairs) runs on `-t` host threads.  Every estimator is restated from the published algorithm the
// crate ports [PARITY UNPINNED, like every crate-internal rule; tools/ref_probe pins them through `<case>.dist.tsv`]:
//   hmh  C / N pair counts (lash_hmh_pair_counts) + LogLog-beta cardinalities + expected-collision correction of
//        axiomhq/hyperminhash (crate hyperminhash 0.1.4);
//   hll  union zero / sum per pair (lash_hll_pair_union_stats) + streaming_algorithms' HLL++ `len()`: linear counting below
//        the published per-precision threshold, else alpha*m^2/sum.  Its third regime (estimate <= 5m: subtract a
//        k-nearest-neighbour bias read from the HLL++ empirical tables) needs data that is not in this image; a sketch
//        or union that falls there is refused with a message instead of being estimated differently;
//   ull  union estimate per pair on the GPU (lash_ull_pair_union_estimates: merged-register histogram + FGRA or ML,
//        ull_estimators.h), per-sketch estimates with lash_ull_estimate; similarity by inclusion-exclusion (utils.rs:272).
// Order: the reference keeps its sketches in hashbrown maps seeded with XXH3(93) and takes the column order, the
// same-files triangle and (under rayon: up to scheduling) the row order from `.keys()`; name_order.hpp
// restates that order, so rows, columns and the (Reference, Query) orientation of each triangle pair come out as the
// reference's `-t 1` run writes them.  --file-order keeps list-file order instead.
// Several GPUs (--devices 0,1,..): blocks of reference rows are handed to one worker per device and written in order.
#include "dist.hpp"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <condition_variable>
#include <map>
#include <mutex>
#include <sstream>
#include <numeric>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/lash_gfx950.h"
#include "dist_format.hpp"
#include "json_out.hpp"
#include "name_order.hpp"
#include "zstd_dl.hpp"

namespace lashhost {
namespace {

// main.rs:284-337
std::string find_files(const std::string &prefix, std::map<std::string, std::string> &out)
{
    std::string norm = prefix;
    size_t slash = norm.find_last_of('/');
    if (slash != std::string::npos) norm = norm.substr(slash + 1);
    if (norm.rfind("./", 0) == 0) norm = norm.substr(2);
    DIR *d = opendir("./");
    if (!d) return "cannot read the current directory";
    out.clear();
    while (dirent *e = readdir(d)) {
        std::string name = e->d_name;
        struct stat st;
        if (stat(name.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
        if (name.rfind(norm, 0) != 0) continue;
        auto ends = [&](const char *suf) { size_t n = strlen(suf); return name.size() >= n && name.compare(name.size() - n, n, suf) == 0; };
        if (ends("parameters.json")) out["params"] = name;
        else if (ends("files.json")) out["files"] = name;
        else if (ends(".bin")) out["sketches"] = name;
    }
    closedir(d);
    if (out.size() != 3) {
        std::ostringstream m;
        m << "There should be 3 files starting with " << norm << " but " << out.size() << " were found instead";
        return m.str();
    }
    return "";
}

std::string slurp(const std::string &path, std::string &out)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) return "cannot open " + path;
    std::ostringstream ss;
    ss << in.rdbuf();
    out = ss.str();
    return "";
}

}  // namespace

std::string run_dist(const DistOptions &opt)
{
    // LASH_CLI_TIMING: where the wall time of a run goes
    const bool timing = getenv("LASH_CLI_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (timing) fprintf(stderr, "[lash dist] %7.3f s  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    std::map<std::string, std::string> rf, qf;
    std::string err = find_files(opt.ref_prefix, rf);
    if (err.empty()) err = find_files(opt.query_prefix, qf);
    if (!err.empty()) return err;
    std::string txt;
    std::map<std::string, std::string> rp, qp;
    if (!(err = slurp(rf["params"], txt)).empty() || !json_parse_string_object(txt, rp)) return err.empty() ? "bad parameters JSON " + rf["params"] : err;
    if (!(err = slurp(qf["params"], txt)).empty() || !json_parse_string_object(txt, qp)) return err.empty() ? "bad parameters JSON " + qf["params"] : err;
    if (rp["k"] != qp["k"]) return "Genomes were not sketched with the same k";                       // main.rs:368-370
    if (rp["algorithm"] != qp["algorithm"]) return "Algorithms do not match in query and sketch genomes";
    const std::string algo = rp["algorithm"];
    if ((algo == "ull" || algo == "hll") && rp["precision"] != qp["precision"])
        return algo + " was not sketched with same precision btwn genomes";
    const int k = atoi(rp["k"].c_str());
    if (opt.model != 0 && opt.model != 1) return "model needs to be 0 or 1";
    const bool hll = algo == "hll", ull = algo == "ull";
    if (!hll && !ull && algo != "hmh") return "Algorithm must be either hmh, ull, or hll";
    int ull_est = LASH_ULL_FGRA;
    if (ull) {                                                                                        // utils.rs:213-217
        if (opt.estimator == "ml") ull_est = LASH_ULL_ML;
        else if (opt.estimator != "fgra") return "estimator needs to be either fgra or ml";
    }
    std::vector<std::string> rnames, qnames;
    if (!(err = slurp(rf["files"], txt)).empty() || !json_parse_string_array(txt, rnames)) return err.empty() ? "bad names JSON " + rf["files"] : err;
    if (!(err = slurp(qf["files"], txt)).empty() || !json_parse_string_array(txt, qnames)) return err.empty() ? "bad names JSON " + qf["files"] : err;
    const bool same_files = qf["files"] == rf["files"];                                               // main.rs:404
    // utils.rs:111-127: the maps' key order (a repeated name is one entry carrying its last sketch)
    std::vector<uint32_t> rorder, qorder;
    if (opt.file_order) {
        rorder.resize(rnames.size()); std::iota(rorder.begin(), rorder.end(), 0u);
        qorder.resize(qnames.size()); std::iota(qorder.begin(), qorder.end(), 0u);
    } else {
        rorder = hashbrown_key_order(rnames);
        qorder = same_files ? rorder : hashbrown_key_order(qnames);
    }

    std::vector<uint8_t> rimg_store, qimg_store;
    if (!(err = zstd_decompress_file(rf["sketches"], rimg_store)).empty()) return err;
    const bool same_sketches = rf["sketches"] == qf["sketches"];                   // all-vs-all: one file, read once
    if (!same_sketches && !(err = zstd_decompress_file(qf["sketches"], qimg_store)).empty()) return err;
    const std::vector<uint8_t> &rimg = rimg_store, &qimg = same_sketches ? rimg_store : qimg_store;
    mark("sketch files read and inflated");
    const int algo_id = hll ? LASH_HLL : ull ? LASH_ULL : LASH_HMH;
    const int prec = (hll || ull) ? atoi(rp["precision"].c_str()) : 0;
    if (hll && (prec < 4 || prec > 16)) return "bad precision in " + rf["params"];
    if (ull && (prec < 3 || prec > 26)) return "bad precision in " + rf["params"];
    const size_t ib = lash_layout_image_bytes(&opt.layout, algo_id, prec);
    if (!ib) return "bad layout";
    if (rimg.size() < rnames.size() * ib) return "Error with reading from " + rf["sketches"];
    if (qimg.size() < qnames.size() * ib) return "Error with reading from " + qf["sketches"];
    // rows / columns: the ENTRIES of the two maps in key order (a repeated name is one entry carrying its last sketch,
    // utils.rs:111-127) — everything below is indexed by position in rorder / qorder, never by name-file index
    const uint32_t nr = (uint32_t)rorder.size(), nq = (uint32_t)qorder.size();
    std::vector<std::string> row_name(nr), col_name(nq);
    for (uint32_t i = 0; i < nr; ++i) row_name[i] = rnames[rorder[i]];
    for (uint32_t j = 0; j < nq; ++j) col_name[j] = qnames[qorder[j]];
    // "q_name == r_name prints 0" (main.rs:452-453) as an integer compare per pair
    std::vector<uint32_t> row_id, col_id;
    name_ids(row_name, col_name, row_id, col_id);
    const std::vector<std::string> col_tab = opt.matrix ? std::vector<std::string>() : tabbed_names(col_name);

    const char *bias_msg = ": cardinality estimate <= 5 * 2^p needs the HLL++ bias tables of streaming_algorithms, which are "
                           "not built in (pass --hll-bias <file from tools/ref_probe/extract_hll_bias.py>, or sketch with a smaller -p)";
    lash_hll_bias *bias = nullptr;
    if (hll && !opt.hll_bias_file.empty()) {
        const int brc = lash_hll_bias_load(opt.hll_bias_file.c_str(), &bias);
        if (brc != LASH_OK) return "cannot read HLL++ bias tables from " + opt.hll_bias_file + ": " + lash_strerror(brc);
    }
    struct BiasGuard { lash_hll_bias *b; ~BiasGuard() { lash_hll_bias_free(b); } } bias_guard{bias};

    // ---- the sketches go to every device ONCE (lash_sketch_set: images + what the pair kernels derive from them), in map order;
    //      per-sketch cardinalities (utils.rs:101-103, 213-217, 314-315) come from register histograms made on the GPU ----
    // Without --devices, two workers share the GPU: while one formats and writes its block the other has the next block's
    // pair statistics computed (a block is GPU work, then -t threads of formatting, then an ordered write).
    std::vector<int> devices = opt.devices.empty() ? std::vector<int>{opt.device, opt.device} : opt.devices;
    struct DevSets { int device = 0; lash_ctx *ctx = nullptr; lash_sketch_set *ref = nullptr, *qry = nullptr; };
    std::vector<DevSets> dev_sets;
    auto free_sets = [&]() {
        for (DevSets &d : dev_sets) {
            if (d.qry && d.qry != d.ref) lash_sketch_set_free(d.ctx, d.qry);
            if (d.ref) lash_sketch_set_free(d.ctx, d.ref);
            if (d.ctx) lash_ctx_destroy(d.ctx);
        }
        dev_sets.clear();
    };
    struct SetsGuard { decltype(free_sets) &f; ~SetsGuard() { f(); } } sets_guard{free_sets};
    const bool one_set = same_sketches && same_files;            // the same images in the same order: one set is both sides
    std::vector<double> rcard(nr), qcard_store;
    for (int dv : devices) {
        bool seen = false;
        for (const DevSets &d : dev_sets) seen = seen || d.device == dv;
        if (seen) continue;
        dev_sets.emplace_back();
        DevSets &d = dev_sets.back();
        d.device = dv;
        int rc = lash_ctx_create(&d.ctx, dv);
        if (rc == LASH_OK) rc = lash_ctx_set_layout(d.ctx, &opt.layout);
        if (rc == LASH_OK) rc = lash_sketch_set_create(d.ctx, algo_id, prec, rimg.data(), (uint32_t)rnames.size(), rorder.data(), nr, &d.ref);
        if (rc == LASH_OK) {
            if (one_set) d.qry = d.ref;
            else rc = lash_sketch_set_create(d.ctx, algo_id, prec, qimg.data(), (uint32_t)qnames.size(), qorder.data(), nq, &d.qry);
        }
        // (every device computes its sets' cardinalities: the sets keep them for the expected-collision vectors)
        uint32_t bad = 0;
        std::vector<double> qc(one_set ? 0 : nq);
        if (rc == LASH_OK) {
            rc = lash_sketch_set_cardinalities(d.ctx, d.ref, ull_est, bias, rcard.data(), &bad);
            if (rc == LASH_ERANGE) return row_name[bad] + bias_msg;
        }
        if (rc == LASH_OK && !one_set) {
            rc = lash_sketch_set_cardinalities(d.ctx, d.qry, ull_est, bias, qc.data(), &bad);
            if (rc == LASH_ERANGE) return col_name[bad] + bias_msg;
            qcard_store = qc;
        }
        if (rc == LASH_OK) rc = lash_sketch_set_prepare(d.ctx, d.ref, d.qry);
        if (rc != LASH_OK) return std::string(lash_strerror(rc)) + " " + lash_ctx_last_error(d.ctx);
    }
    const std::vector<double> &qcard = one_set ? rcard : qcard_store;
    mark("sketches resident on the device(s), cardinalities, pair-kernel operands");
    // hyperminhash's expected collisions need the GPU only when some pair has both sketches at or below 2^19 distinct k-mers
    bool small_ref = false, small_qry = false;
    if (!hll && !ull) {
        for (double c : rcard) small_ref = small_ref || !(c > 524288.0);
        for (double c : qcard) small_qry = small_qry || !(c > 524288.0);
    }
    const bool gpu_ec = small_ref && small_qry;
    FILE *out = fopen(opt.output_file.c_str(), "w");
    if (!out) return "cannot create " + opt.output_file;
    if (!opt.matrix) fprintf(out, "Reference\tQuery\tDistance\n");                                   // main.rs:409-412
    else for (uint32_t j = 0; j < nq; ++j) fprintf(out, "\t%s", col_name[j].c_str());                // main.rs:439-441
    // ---- blocks of reference rows: bounded pair tables (all-vs-all on 10^5 sketches is 5 * 10^9 printed pairs), about the
    //      same number of PRINTED pairs each — with same files row i prints i + 1 columns, so late blocks hold fewer rows ----
    std::vector<uint32_t> block_begin{0};
    {
        const uint64_t total = same_files ? (uint64_t)nr * (nr + 1) / 2 : (uint64_t)nr * nq;
        const uint64_t want = opt.block_rows ? 0 : std::max<uint64_t>(std::min<uint64_t>(32ull << 20, total / 16 + 1), 4096);   // pairs per block, >= ~16 blocks
        uint64_t acc = 0;
        for (uint32_t i = 0; i < nr; ++i) {
            acc += same_files ? i + 1 : nq;
            const bool cut = opt.block_rows ? (i + 1 - block_begin.back()) >= opt.block_rows : acc >= want;
            if (cut && i + 1 < nr) { block_begin.push_back(i + 1); acc = 0; }
        }
        if (nr) block_begin.push_back(nr);
    }
    const uint32_t n_blocks = (uint32_t)block_begin.size() - 1;
    if (devices.size() > n_blocks) devices.resize(std::max<uint32_t>(n_blocks, 1));
    const int fmt_threads = std::max(1, opt.threads / (int)devices.size());
    std::atomic<uint32_t> next_block{0};
    std::mutex wmu;
    std::condition_variable wcv;
    uint32_t next_to_write = 0;
    std::string fail;                                            // guarded by wmu

    auto worker = [&](int device) {
        const DevSets *ds = nullptr;
        for (const DevSets &d : dev_sets) if (d.device == device) ds = &d;
        lash_ctx *ctx = nullptr;
        int rc = lash_ctx_create(&ctx, device);                  // the worker's own stream and staging; the sets are shared, read-only
        if (rc == LASH_OK) rc = lash_ctx_set_layout(ctx, &opt.layout);
        std::string my_fail = rc == LASH_OK ? "" : std::string(lash_strerror(rc));
        // pair tables in page-locked memory (the copy back runs at the link rate): hmh C / N; hll zero + sum; ull the union estimate
        uint32_t *C = nullptr, *N = nullptr;
        double *U = nullptr, *EC = nullptr;
        size_t cap = 0, ec_cap = 0;
        RowText row_text;                                        // the block's text; its memory is reused from block to block
        auto grow = [&](size_t np) {
            if (np <= cap) return true;
            lash_host_free_pinned(C); lash_host_free_pinned(N); lash_host_free_pinned(U);
            C = N = nullptr; U = nullptr;
            cap = np + np / 8;
            if (!ull) C = static_cast<uint32_t *>(lash_host_alloc_pinned(cap * 4));
            if (!hll && !ull) N = static_cast<uint32_t *>(lash_host_alloc_pinned(cap * 4));
            if (hll || ull) U = static_cast<double *>(lash_host_alloc_pinned(cap * 8));
            return (ull || C) && (hll || ull || N) && (!(hll || ull) || U);
        };
        for (;;) {
            const uint32_t blk = next_block.fetch_add(1);
            if (blk >= n_blocks) break;
            const uint32_t i0 = block_begin[blk], i1 = block_begin[blk + 1];
            const uint32_t n_cols = same_files ? std::min(i1, nq) : nq;     // the triangle: no row of the block prints beyond its own column
            bool skip;
            { std::lock_guard<std::mutex> lk(wmu); skip = !fail.empty(); }
            bool have_ec = false;
            row_text.off.clear(); row_text.len.clear();
            if (my_fail.empty() && !skip) {
                const size_t np = (size_t)(i1 - i0) * n_cols;
                if (!grow(np)) my_fail = "out of page-locked host memory";
                else {
                    rc = lash_sketch_set_pair_block(ctx, ds->ref, i0, i1, ds->qry, n_cols, same_files ? 1 : 0, ull_est, C, N, U);
                    if (rc == LASH_OK && gpu_ec) {
                        // hyperminhash's expected_collisions below 2^19 distinct k-mers on both sides is a 65 536-cell sum — on the host
                        // 4 ms to 0.2 s per pair; the library does the block's small pairs as one matrix product on the GPU
                        if (np > ec_cap) { lash_host_free_pinned(EC); ec_cap = np + np / 8; EC = static_cast<double *>(lash_host_alloc_pinned(ec_cap * 8)); }
                        uint64_t n_small = 0;
                        rc = EC ? lash_sketch_set_hmh_expected_collisions(ctx, ds->ref, i0, i1, ds->qry, n_cols, EC, &n_small) : LASH_ENOMEM;
                        have_ec = n_small != 0;
                    }
                    if (rc != LASH_OK) my_fail = std::string(lash_strerror(rc)) + " " + lash_ctx_last_error(ctx);
                }
            }
            if (my_fail.empty() && !skip) {
                BlockTables bt;
                bt.c_or_zero = C; bt.n_counts = N; bt.sum_or_union = U; bt.hmh_ec = have_ec ? EC : nullptr; bt.ld = n_cols;
                my_fail = dist_block_rows(algo_id, prec, k, opt.model, opt.fp32, bias, i0, i1, same_files, nq, rcard.data(), qcard.data(), bt, row_name,
                                          col_name, col_tab, row_id.data(), col_id.data(), opt.matrix, fmt_threads, row_text);
                if (!my_fail.empty()) row_text.len.clear();
            }
            // in block order, whatever order the devices finish in
            std::unique_lock<std::mutex> lk(wmu);
            wcv.wait(lk, [&] { return next_to_write == blk; });
            if (!my_fail.empty() && fail.empty()) fail = my_fail;
            if (fail.empty())
                for (size_t r = 0; r < row_text.rows(); ++r) fwrite(row_text.data(r), 1, row_text.size(r), out);
            ++next_to_write;
            lk.unlock();
            wcv.notify_all();
        }
        lash_host_free_pinned(C); lash_host_free_pinned(N); lash_host_free_pinned(U); lash_host_free_pinned(EC);
        if (ctx) lash_ctx_destroy(ctx);
    };
    {
        std::vector<std::thread> pool;
        for (size_t d = 1; d < devices.size(); ++d) pool.emplace_back(worker, devices[d]);
        worker(devices[0]);
        for (auto &t : pool) t.join();
    }
    fclose(out);
    mark("all rows written");
    return fail;
}

}  // namespace lashhost
