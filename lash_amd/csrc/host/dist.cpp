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
    std::unordered_map<std::string, uint32_t> qpos;                                                   // utils.rs:130-142 file_idx
    if (same_files && !opt.file_order)
        for (uint32_t jj = 0; jj < qorder.size(); ++jj) qpos[qnames[qorder[jj]]] = jj;

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
    const size_t ib = lash_layout_image_bytes(&opt.layout, algo_id, prec), hdr = lash_layout_header_bytes(&opt.layout, algo_id);
    if (!ib) return "bad layout";
    if (rimg.size() < rnames.size() * ib) return "Error with reading from " + rf["sketches"];
    if (qimg.size() < qnames.size() * ib) return "Error with reading from " + qf["sketches"];
    const uint32_t nr = (uint32_t)rorder.size(), nq = (uint32_t)qnames.size();   // rows: map entries; columns of the pair tables: every query image

    std::vector<double> rcard(nr), qcard(nq);
    const char *bias_msg = ": cardinality estimate <= 5 * 2^p needs the HLL++ bias tables of streaming_algorithms, which are "
                           "not built in (pass --hll-bias <file from tools/ref_probe/extract_hll_bias.py>, or sketch with a smaller -p)";
    lash_hll_bias *bias = nullptr;
    if (hll && !opt.hll_bias_file.empty()) {
        const int brc = lash_hll_bias_load(opt.hll_bias_file.c_str(), &bias);
        if (brc != LASH_OK) return "cannot read HLL++ bias tables from " + opt.hll_bias_file + ": " + lash_strerror(brc);
    }
    struct BiasGuard { lash_hll_bias *b; ~BiasGuard() { lash_hll_bias_free(b); } } bias_guard{bias};
    // per-sketch cardinalities (utils.rs:101-103, 213-217, 314-315), on `-t` host threads
    auto cards = [&](const std::vector<uint8_t> &img, const std::vector<std::string> &names, std::vector<double> &card) -> std::string {
        const uint32_t n = (uint32_t)names.size();
        std::vector<uint8_t> bad(n, 0);
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (uint32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
                const uint8_t *regs = img.data() + (size_t)i * ib + hdr;
                if (ull) card[i] = lash_ull_estimate(regs, prec, ull_est);
                else if (!hll) card[i] = lash_hmh_cardinality(regs, opt.layout.hmh_reg_be != 0);
                else if (lash_hll_cardinality(regs, prec, bias, &card[i]) != LASH_OK) bad[i] = 1;
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < std::min<int>(opt.threads, (int)n); ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        for (uint32_t i = 0; i < n; ++i)
            if (bad[i]) return names[i] + bias_msg;
        return "";
    };
    if (!(err = cards(rimg, rnames, rcard)).empty()) return err;
    if (same_files && rf["sketches"] == qf["sketches"]) qcard = rcard;
    else if (!(err = cards(qimg, qnames, qcard)).empty()) return err;

    mark("per-sketch cardinalities");
    // hyperminhash's expected collisions need the GPU only when some pair has both sketches at or below 2^19 distinct k-mers
    bool small_ref = false, small_qry = false;
    if (!hll && !ull) {
        for (uint32_t i : rorder) small_ref = small_ref || !(rcard[i] > 524288.0);
        for (double c : qcard) small_qry = small_qry || !(c > 524288.0);
    }
    const bool gpu_ec = small_ref && small_qry;
    FILE *out = fopen(opt.output_file.c_str(), "w");
    if (!out) return "cannot create " + opt.output_file;
    if (!opt.matrix) fprintf(out, "Reference\tQuery\tDistance\n");                                   // main.rs:409-412
    else for (uint32_t j : qorder) fprintf(out, "\t%s", qnames[j].c_str());                          // main.rs:439-441
    // ---- GPU: the O(N_ref * N_qry * registers) scan, in blocks of reference rows so that the per-pair tables stay
    //      bounded (all-vs-all on 10^5 sketches is 10^10 pairs); one worker (context + host thread) per device ----
    // Without --devices, two workers share the GPU: while one formats and writes its block the other has the next block's
    // pair statistics computed (a block is GPU work, then -t threads of formatting, then an ordered write).
    std::vector<int> devices = opt.devices.empty() ? std::vector<int>{opt.device, opt.device} : opt.devices;
    // rows per block: pair tables of at most 64 M entries, and at least ~16 blocks so that the workers overlap
    const uint32_t rows_per_block = opt.block_rows ? std::min(opt.block_rows, std::max(nr, 1u))
                                                   : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint64_t>(64, (nr + 15) / 16),
                                                                                                       std::min<uint64_t>(nr, (64ull << 20) / std::max<uint32_t>(nq, 1))));
    const uint32_t n_blocks = nr ? (nr + rows_per_block - 1) / rows_per_block : 0;
    if (devices.size() > n_blocks) devices.resize(std::max<uint32_t>(n_blocks, 1));
    const int fmt_threads = std::max(1, opt.threads / (int)devices.size());
    std::atomic<uint32_t> next_block{0};
    std::mutex wmu;
    std::condition_variable wcv;
    uint32_t next_to_write = 0;
    std::string fail;                                            // guarded by wmu

    auto worker = [&](int device) {
        lash_ctx *ctx = nullptr;
        int rc = lash_ctx_create(&ctx, device);
        if (rc == LASH_OK) rc = lash_ctx_set_layout(ctx, &opt.layout);
        std::string my_fail = rc == LASH_OK ? "" : std::string(lash_strerror(rc));
        std::vector<uint32_t> C, N;                              // hmh: C / N; hll: C = zero registers of the union
        std::vector<double> U;                                   // hll: union sum; ull: union estimate
        std::vector<double> EC, rc_blk;                          // hmh: expected collisions of the block's pairs (GPU), its row cardinalities
        std::vector<uint8_t> gathered;                           // the block's reference images when rows are not in file order
        for (;;) {
            const uint32_t blk = next_block.fetch_add(1);
            if (blk >= n_blocks) break;
            const uint32_t i0 = blk * rows_per_block, i1 = std::min(nr, i0 + rows_per_block);
            std::vector<std::string> row_text(i1 - i0), row_fail(i1 - i0);
            bool skip;
            { std::lock_guard<std::mutex> lk(wmu); skip = !fail.empty(); }
            if (my_fail.empty() && !skip) {
                const size_t np = (size_t)(i1 - i0) * nq;
                if (!ull) C.resize(np);
                if (!hll && !ull) N.resize(np);
                if (hll || ull) U.resize(np);
                const uint8_t *rblk = rimg.data() + (size_t)i0 * ib;
                if (!opt.file_order) {
                    gathered.resize((size_t)(i1 - i0) * ib);
                    for (uint32_t i = i0; i < i1; ++i) memcpy(gathered.data() + (size_t)(i - i0) * ib, rimg.data() + (size_t)rorder[i] * ib, ib);
                    rblk = gathered.data();
                }
                rc = hll ? lash_hll_pair_union_stats(ctx, prec, rblk, i1 - i0, qimg.data(), nq, C.data(), U.data())
                   : ull ? lash_ull_pair_union_estimates(ctx, prec, ull_est, rblk, i1 - i0, qimg.data(), nq, U.data())
                         : lash_hmh_pair_counts(ctx, rblk, i1 - i0, qimg.data(), nq, C.data(), N.data());
                if (rc == LASH_OK && gpu_ec) {
                    // hyperminhash's expected_collisions for the block: O(1) per pair above 2^19 distinct k-mers, a 65 536-cell sum
                    // below — on the host that is 4 ms to 0.2 s per pair; the library does it as one matrix product on the GPU
                    rc_blk.resize(i1 - i0);
                    for (uint32_t i = i0; i < i1; ++i) rc_blk[i - i0] = rcard[rorder[i]];
                    EC.resize(np);
                    rc = lash_hmh_pair_expected_collisions(ctx, rc_blk.data(), i1 - i0, qcard.data(), nq, EC.data());
                }
                if (rc != LASH_OK) my_fail = std::string(lash_strerror(rc)) + " " + lash_ctx_last_error(ctx);
            }
            if (my_fail.empty() && !skip) {
                // rows of the block are formatted by host threads (the reference's par_iter over reference sketches,
                // utils.rs:146,248,342), then written in file order
                auto do_row = [&](uint32_t i) {
                    std::string &text = row_text[i - i0];
                    char buf[64];
                    bool first = true;
                    const size_t row = (size_t)(i - i0) * nq;
                    std::vector<double> dist(nq);
                    uint64_t bad_pair = 0;
                    const uint32_t ri = rorder[i];
                    const std::string &rname = rnames[ri];
                    const uint32_t my_pos = !same_files ? 0 : opt.file_order ? i : qpos.at(rname);
                    const int drc = lash_dist_rows(algo_id, prec, k, opt.model, opt.fp32 ? 1 : 0, 1, nq, &rcard[ri], qcard.data(),
                                                   ull ? nullptr : C.data() + row, (hll || ull) ? nullptr : N.data() + row,
                                                   (hll || ull) ? U.data() + row : nullptr, bias, gpu_ec ? EC.data() + row : nullptr,
                                                   dist.data(), &bad_pair);
                    if (drc == LASH_ERANGE) { row_fail[i - i0] = "union of " + rname + " and " + qnames[bad_pair] + bias_msg; return; }
                    if (drc != LASH_OK) { row_fail[i - i0] = lash_strerror(drc); return; }
                    for (uint32_t jj = 0; jj < qorder.size(); ++jj) {
                        if (same_files && jj > my_pos) continue;                                      // utils.rs:158-160
                        const uint32_t j = qorder[jj];
                        const double d = qnames[j] == rname ? 0.0 : dist[j];                          // main.rs:452-453
                        // "{:.6}" (main.rs:456,461): std::to_chars is correctly rounded like printf("%.6f") and several times faster
                        buf[0] = '\t';
                        char *end = std::to_chars(buf + 1, buf + sizeof buf - 2, d, std::chars_format::fixed, 6).ptr;
                        if (!opt.matrix) {
                            text += rname; text += '\t'; text += qnames[j];
                            *end++ = '\n';
                        } else if (first) { text += '\n'; text += rname; }
                        text.append(buf, end);
                        first = false;
                    }
                };
                const uint32_t nthreads = (uint32_t)std::max(1, std::min<int>(fmt_threads, (int)(i1 - i0)));
                std::atomic<uint32_t> next{i0};
                std::vector<std::thread> pool;
                auto work = [&]() { for (uint32_t i = next.fetch_add(1); i < i1; i = next.fetch_add(1)) do_row(i); };
                for (uint32_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
                work();
                for (auto &t : pool) t.join();
                for (const std::string &f : row_fail) if (!f.empty() && my_fail.empty()) my_fail = f;
            }
            // in block order, whatever order the devices finish in
            std::unique_lock<std::mutex> lk(wmu);
            wcv.wait(lk, [&] { return next_to_write == blk; });
            if (!my_fail.empty() && fail.empty()) fail = my_fail;
            if (fail.empty())
                for (const std::string &t : row_text) fwrite(t.data(), 1, t.size(), out);
            ++next_to_write;
            lk.unlock();
            wcv.notify_all();
        }
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
