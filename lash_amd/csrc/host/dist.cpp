// dist.cpp — `lash dist` for the gfx950 build.
//
// Mirrors /root/reference/src/main.rs:280-617 (file discovery, parameter checks, output formats, distance formula)
// and utils.rs:84-184 (hmh_distance).  The O(N_ref * N_qry * 16384) register scan runs on the GPU
// (lash_hmh_pair_counts); cardinalities (LogLog-beta) and the expected-collision correction are restated from
// axiomhq/hyperminhash, which the crate hyperminhash 0.1.4 ports [PARITY UNPINNED, like every crate-internal rule].
// Row order: the reference iterates hashbrown maps under rayon (nondeterministic); here rows come in
// file order, so parity with the reference is on the SET of rows.
// HyperLogLog (utils.rs:290-373): per pair union -> len -> inclusion-exclusion Jaccard.  The union's zero / sum come
// from the GPU (lash_hll_pair_union_stats); `len()` is streaming_algorithms' HLL++ estimator restated [PARITY
// UNPINNED]: linear counting below the published per-precision threshold, else alpha*m^2/sum.  Its third regime
// (estimate <= 5m: subtract a k-nearest-neighbour bias read from the HLL++ empirical tables) needs data that is not
// in this image; a sketch or union that falls there is refused with a message instead of being estimated differently.
// UltraLogLog distances need the FGRA / ML estimator constants of ultraloglog 0.1.6 and are not built.
#include "dist.hpp"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>
#include <vector>

#include "../../../include/lash_gfx950.h"
#include "json_out.hpp"
#include "zstd_dl.hpp"

namespace lashhost {
namespace {

constexpr int HP = 14, HQ = 6, HR = 10;
constexpr uint32_t HM = 1u << HP;

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

// hyperminhash (axiomhq) restated: LogLog-beta cardinality of one sketch
double hmh_beta(double ez)
{
    const double zl = std::log(ez + 1.0);
    return -0.370393911 * ez + 0.070471823 * zl + 0.17393686 * std::pow(zl, 2) + 0.16339839 * std::pow(zl, 3) +
           -0.09237745 * std::pow(zl, 4) + 0.03738027 * std::pow(zl, 5) + -0.005384159 * std::pow(zl, 6) +
           0.00042419 * std::pow(zl, 7);
}

double hmh_cardinality(const uint8_t *img)
{
    double sum = 0.0, ez = 0.0;
    for (uint32_t i = 0; i < HM; ++i) {
        const uint32_t reg = img[2 * i] | (img[2 * i + 1] << 8);
        const uint32_t lz = reg >> (16 - HQ);
        if (lz == 0) ez += 1.0;
        sum += std::ldexp(1.0, -(int)lz);                       // == 1 / 2^lz exactly
    }
    const double m = (double)HM;
    const double alpha = 0.7213 / (1.0 + 1.079 / m);
    return alpha * m * (m - ez) / (hmh_beta(ez) + sum);
}

double hmh_expected_collision(double n, double m)
{
    const double two_q = 64.0, two_r = 1024.0;
    double x = 0.0;
    for (double i = 1.0; i <= two_q; i += 1.0) {
        for (double j = 1.0; j <= two_r; j += 1.0) {
            double b1, b2;
            if (i != two_q) {
                const double den = std::pow(2.0, HP + HR + i);
                b1 = (two_r + j) / den;
                b2 = (two_r + j + 1.0) / den;
            } else {
                const double den = std::pow(2.0, HP + HR + i - 1.0);
                b1 = j / den;
                b2 = (j + 1.0) / den;
            }
            const double prx = std::pow(1.0 - b2, n) - std::pow(1.0 - b1, n);
            const double pry = std::pow(1.0 - b2, m) - std::pow(1.0 - b1, m);
            x += prx * pry;
        }
    }
    return x * (double)HP + 0.5;
}

double hmh_approx_expected_collisions(double n, double m)
{
    if (n < m) std::swap(n, m);
    if (n > std::pow(2.0, std::pow(2.0, (double)HQ) + (double)HR)) return 1.8446744073709552e19;   // u64::MAX
    if (n > std::pow(2.0, (double)(HP + 5))) {
        const double d = (4.0 * n / m) / std::pow((1.0 + n) / m, 2.0);
        return 0.169919487159739093975315012348 * std::pow(2.0, (double)(HP - HR)) * d + 0.5;
    }
    return hmh_expected_collision(n, m) / (double)HP;
}

// ---- HyperLogLog len() (streaming_algorithms 0.3.3, HLL++ as published by Heule et al.) ----
constexpr double HLL_THRESHOLD[15] = {10, 20, 40, 80, 220, 400, 900, 1800, 3100, 6500, 11500, 20000, 50000, 120000, 350000};  // p = 4..18

// returns false when the estimate falls in the bias-corrected regime (no tables here)
bool hll_len(int p, double alpha, uint64_t zero, double sum, double &out)
{
    const double m = (double)(1u << p);
    if (zero > 0) {
        const double h = m * std::log(m / (double)zero);
        if (h <= HLL_THRESHOLD[p - 4]) { out = h; return true; }
    }
    const double e = alpha * m * m / sum;
    if (e <= 5.0 * m) return false;
    out = e;
    return true;
}

uint64_t rd_u64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
double rd_f64(const uint8_t *p) { double v; memcpy(&v, p, 8); return v; }

template <class T>
T compute_distance(T frac, int k, int model)
{
    const T kk = (T)k;
    if (model == 1) { const T d = -std::log(frac) / kk; return d < (T)1 ? d : (T)1; }      // (-frac.ln() / k).min(1)
    return (T)1 - std::pow(frac, (T)1 / kk);
}

}  // namespace

std::string run_dist(const DistOptions &opt)
{
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
    std::vector<std::string> rnames, qnames;
    if (!(err = slurp(rf["files"], txt)).empty() || !json_parse_string_array(txt, rnames)) return err.empty() ? "bad names JSON " + rf["files"] : err;
    if (!(err = slurp(qf["files"], txt)).empty() || !json_parse_string_array(txt, qnames)) return err.empty() ? "bad names JSON " + qf["files"] : err;
    if (algo == "ull")
        return "lash dist for -a ull is not built in the gfx950 port yet (it needs the ultraloglog FGRA/ML estimator constants)";
    const bool same_files = qf["files"] == rf["files"];                                               // main.rs:404

    std::vector<uint8_t> rimg, qimg;
    if (!(err = zstd_decompress_file(rf["sketches"], rimg)).empty()) return err;
    if (!(err = zstd_decompress_file(qf["sketches"], qimg)).empty()) return err;
    const bool hll = algo == "hll";
    const int prec = hll ? atoi(rp["precision"].c_str()) : 0;
    if (hll && (prec < 4 || prec > 16)) return "bad precision in " + rf["params"];
    const size_t ib = hll ? lash_sketch_image_bytes(LASH_HLL, prec) : (size_t)HM * 2;
    if (rimg.size() < rnames.size() * ib) return "Error with reading from " + rf["sketches"];
    if (qimg.size() < qnames.size() * ib) return "Error with reading from " + qf["sketches"];
    const uint32_t nr = (uint32_t)rnames.size(), nq = (uint32_t)qnames.size();

    std::vector<double> rcard(nr), qcard(nq);
    const char *bias_msg = ": cardinality estimate <= 5 * 2^p needs the HLL++ bias tables of streaming_algorithms, which "
                           "this build does not have (sketch with a smaller -p)";
    // per-sketch cardinalities (utils.rs:101-103, 314-315), on `-t` host threads
    auto cards = [&](const std::vector<uint8_t> &img, const std::vector<std::string> &names, std::vector<double> &card) -> std::string {
        const uint32_t n = (uint32_t)names.size();
        std::vector<uint8_t> bad(n, 0);
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (uint32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
                const uint8_t *im = img.data() + (size_t)i * ib;
                if (!hll) card[i] = hmh_cardinality(im);
                else if (!hll_len(prec, rd_f64(im), rd_u64(im + 8), rd_f64(im + 16), card[i])) bad[i] = 1;
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
    const double hll_alpha = hll && nr ? rd_f64(rimg.data()) : 0.0;

    FILE *out = fopen(opt.output_file.c_str(), "w");
    if (!out) return "cannot create " + opt.output_file;
    if (!opt.matrix) fprintf(out, "Reference\tQuery\tDistance\n");                                   // main.rs:409-412
    else for (uint32_t j = 0; j < nq; ++j) fprintf(out, "\t%s", qnames[j].c_str());                   // main.rs:439-441
    // ---- GPU: the O(N_ref * N_qry * registers) scan, in blocks of reference rows so that the per-pair tables stay
    //      bounded (all-vs-all on 10^5 sketches is 10^10 pairs) ----
    lash_ctx *ctx = nullptr;
    {
        const int rc = lash_ctx_create(&ctx, opt.device);
        if (rc != LASH_OK) { fclose(out); return lash_strerror(rc); }
    }
    const uint32_t rows_per_block = opt.block_rows ? std::min(opt.block_rows, std::max(nr, 1u))
                                                   : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nr, (64ull << 20) / std::max<uint32_t>(nq, 1)));
    std::vector<uint32_t> C((size_t)rows_per_block * nq), N(hll ? 0 : (size_t)rows_per_block * nq);   // hll: C = zero registers of the union
    std::vector<double> usum(hll ? (size_t)rows_per_block * nq : 0);
    std::string fail;
    for (uint32_t i0 = 0; i0 < nr && fail.empty(); i0 += rows_per_block) {
        const uint32_t i1 = std::min(nr, i0 + rows_per_block);
        const int rc = hll ? lash_hll_pair_union_stats(ctx, prec, rimg.data() + (size_t)i0 * ib, i1 - i0, qimg.data(), nq, C.data(), usum.data())
                           : lash_hmh_pair_counts(ctx, rimg.data() + (size_t)i0 * ib, i1 - i0, qimg.data(), nq, C.data(), N.data());
        if (rc != LASH_OK) { fail = std::string(lash_strerror(rc)) + " " + lash_ctx_last_error(ctx); break; }
        // rows of the block are formatted by `-t` host threads (the reference's par_iter over reference sketches,
        // utils.rs:146,336), then written in file order
        std::vector<std::string> row_text(i1 - i0), row_fail(i1 - i0);
        auto do_row = [&](uint32_t i) {
            std::string &txt = row_text[i - i0];
            char buf[64];
            bool first = true;
            const size_t row = (size_t)(i - i0) * nq;
            for (uint32_t j = 0; j < nq; ++j) {
                if (same_files && j > i) continue;                                                    // utils.rs:158-160
                double sim = 0.0;
                if (hll) {                                                                            // utils.rs:352-365
                    double u;
                    if (!hll_len(prec, hll_alpha, C[row + j], usum[row + j], u)) {
                        row_fail[i - i0] = "union of " + rnames[i] + " and " + qnames[j] + bias_msg;
                        return;
                    }
                    sim = (rcard[i] + qcard[j] - u) / u;
                } else {
                    const double c = (double)C[row + j], n = (double)N[row + j];
                    if (c != 0.0) {                                                                   // Sketch::similarity
                        const double ec = hmh_approx_expected_collisions(qcard[j], rcard[i]);
                        sim = c < ec ? 0.0 : (c - ec) / n;
                    }
                }
                if (sim < 0.0) sim = 0.0;                                                             // .max(0.0), utils.rs:164,362
                const double frac = 2.0 * sim / (1.0 + sim);                                          // utils.rs:165-167
                double d;
                if (qnames[j] == rnames[i]) d = 0.0;                                                  // main.rs:452-453
                else if (opt.fp32) d = (double)compute_distance<float>((float)frac, k, opt.model);
                else d = compute_distance<double>(frac, k, opt.model);
                // "{:.6}" (main.rs:456,461): std::to_chars is correctly rounded like printf("%.6f") and several times faster
                buf[0] = '\t';
                char *end = std::to_chars(buf + 1, buf + sizeof buf - 2, d, std::chars_format::fixed, 6).ptr;
                if (!opt.matrix) {
                    txt += rnames[i]; txt += '\t'; txt += qnames[j];
                    *end++ = '\n';
                } else if (first) { txt += '\n'; txt += rnames[i]; }
                txt.append(buf, end);
                first = false;
            }
        };
        {
            const uint32_t nthreads = (uint32_t)std::max(1, std::min<int>(opt.threads, (int)(i1 - i0)));
            std::atomic<uint32_t> next{i0};
            std::vector<std::thread> pool;
            auto work = [&]() { for (uint32_t i = next.fetch_add(1); i < i1; i = next.fetch_add(1)) do_row(i); };
            for (uint32_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
            work();
            for (auto &t : pool) t.join();
        }
        for (uint32_t i = i0; i < i1; ++i) {
            if (!row_fail[i - i0].empty()) { fail = row_fail[i - i0]; break; }
            fwrite(row_text[i - i0].data(), 1, row_text[i - i0].size(), out);
        }
    }
    lash_ctx_destroy(ctx);
    if (!fail.empty()) { fclose(out); return fail; }
    fclose(out);
    return "";
}

}  // namespace lashhost
