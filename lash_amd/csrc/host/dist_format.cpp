// dist_format.cpp — see dist_format.hpp.
#include "dist_format.hpp"

#include <charconv>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <unordered_map>

#include "../../../include/lash_gfx950.h"

namespace lashhost {

std::vector<std::string> tabbed_names(const std::vector<std::string> &names)
{
    std::vector<std::string> t(names.size());
    for (size_t i = 0; i < names.size(); ++i) { t[i].reserve(names[i].size() + 2); t[i] += '\t'; t[i] += names[i]; t[i] += '\t'; }
    return t;
}

void name_ids(const std::vector<std::string> &row_name, const std::vector<std::string> &col_name, std::vector<uint32_t> &row_id,
              std::vector<uint32_t> &col_id)
{
    std::unordered_map<std::string, uint32_t> ids;
    ids.reserve(row_name.size() + col_name.size());
    auto id_of = [&](const std::string &s) { return ids.emplace(s, (uint32_t)ids.size()).first->second; };
    row_id.resize(row_name.size());
    col_id.resize(col_name.size());
    for (size_t i = 0; i < row_name.size(); ++i) row_id[i] = id_of(row_name[i]);
    for (size_t j = 0; j < col_name.size(); ++j) col_id[j] = id_of(col_name[j]);
}

size_t row_text_bound(const std::string &rname, const std::vector<std::string> &qtab, uint32_t n_print, bool matrix)
{
    if (matrix) return 1 + rname.size() + (size_t)n_print * 28;
    size_t qbytes = 0;
    for (uint32_t c = 0; c < n_print; ++c) qbytes += qtab[c].size();
    return (size_t)n_print * (rname.size() + 28) + qbytes;
}

size_t format_row(char *dst, const std::string &rname, const std::vector<std::string> &qtab, uint32_t n_print, const double *dist, uint32_t row_id,
                  const uint32_t *col_id, bool matrix)
{
    if (n_print == 0) return 0;                                                  // (a row with no column prints nothing, main.rs:443)
    // "{:.6}" (main.rs:456,461): std::to_chars(fixed, 6) is correctly rounded like Rust's formatter, several times faster than printf
    char *p = dst;
    if (matrix) { *p++ = '\n'; memcpy(p, rname.data(), rname.size()); p += rname.size(); }
    for (uint32_t c = 0; c < n_print; ++c) {
        const double d = col_id[c] == row_id ? 0.0 : dist[c];
        if (matrix) *p++ = '\t';
        else {
            memcpy(p, rname.data(), rname.size()); p += rname.size();
            memcpy(p, qtab[c].data(), qtab[c].size()); p += qtab[c].size();
        }
        if (d == 1.0) { memcpy(p, "1.000000", 8); p += 8; }                      // unrelated genomes: most of an all-vs-all
        else if (d == 0.0) { memcpy(p, "0.000000", 8); p += 8; }
        else if (d != d) { memcpy(p, "NaN", 3); p += 3; }                         // Rust's Display (ull, model 0, two empty sketches)
        else p = std::to_chars(p, p + 26, d, std::chars_format::fixed, 6).ptr;
        if (!matrix) *p++ = '\n';
    }
    return (size_t)(p - dst);
}

void append_row(std::string &out, const std::string &rname, const std::vector<std::string> &qtab, uint32_t n_print, const double *dist,
                uint32_t row_id, const uint32_t *col_id, bool matrix)
{
    if (n_print == 0) return;
    const size_t at0 = out.size();
    out.resize(at0 + row_text_bound(rname, qtab, n_print, matrix));
    out.resize(at0 + format_row(&out[at0], rname, qtab, n_print, dist, row_id, col_id, matrix));
}

std::string dist_block_rows(int algo, int p, int k, int model, bool fp32, const void *hll_bias, uint32_t i0, uint32_t i1, bool triangle,
                            uint32_t n_cols_total, const double *row_card, const double *col_card, const BlockTables &t,
                            const std::vector<std::string> &row_name, const std::vector<std::string> &col_name,
                            const std::vector<std::string> &col_tab, const uint32_t *row_id, const uint32_t *col_id, bool matrix, int threads,
                            RowText &text)
{
    static const char *bias_msg = ": cardinality estimate <= 5 * 2^p needs the HLL++ bias tables of streaming_algorithms, which are "
                                  "not built in (pass --hll-bias <file from tools/ref_probe/extract_hll_bias.py>, or sketch with a smaller -p)";
    const bool hll = algo == LASH_HLL, ull = algo == LASH_ULL;
    const uint32_t n_rows = i1 - i0;
    // slots: row r at off[r], room for its upper bound (list form: the column names' bytes come from a running sum)
    text.off.assign(n_rows, 0);
    text.len.assign(n_rows, 0);
    {
        std::vector<size_t> qsum;                                                 // qsum[c] = bytes of col_tab[0 .. c)
        if (!matrix) { qsum.resize((size_t)n_cols_total + 1, 0); for (uint32_t c = 0; c < n_cols_total; ++c) qsum[c + 1] = qsum[c] + col_tab[c].size(); }
        size_t at = 0;
        for (uint32_t r = 0; r < n_rows; ++r) {
            const uint32_t i = i0 + r, n_print = triangle ? std::min(i + 1, n_cols_total) : n_cols_total;
            text.off[r] = at;
            at += matrix ? 1 + row_name[i].size() + (size_t)n_print * 28 : (size_t)n_print * (row_name[i].size() + 28) + qsum[n_print];
        }
        if (text.buf.size() < at) text.buf.resize(at + at / 8);
    }
    std::vector<std::string> row_fail(n_rows);
    auto do_row = [&](uint32_t i, std::vector<double> &dist) {
        const size_t row = (size_t)(i - i0) * t.ld;
        const uint32_t n_print = triangle ? std::min(i + 1, n_cols_total) : n_cols_total;                  // utils.rs:158-160
        if (dist.size() < n_print) dist.resize(n_print);
        uint64_t bad_pair = 0;
        const int drc = lash_dist_rows(algo, p, k, model, fp32 ? 1 : 0, 1, n_print, &row_card[i], col_card, ull ? nullptr : t.c_or_zero + row,
                                       (hll || ull) ? nullptr : t.n_counts + row, (hll || ull) ? t.sum_or_union + row : nullptr,
                                       static_cast<const lash_hll_bias *>(hll_bias), t.hmh_ec ? t.hmh_ec + row : nullptr, dist.data(), &bad_pair);
        if (drc == LASH_ERANGE) { row_fail[i - i0] = "union of " + row_name[i] + " and " + col_name[bad_pair] + bias_msg; return; }
        if (drc != LASH_OK) { row_fail[i - i0] = lash_strerror(drc); return; }
        text.len[i - i0] = format_row(text.buf.data() + text.off[i - i0], row_name[i], col_tab, n_print, dist.data(), row_id[i], col_id, matrix);
    };
    const uint32_t nthreads = (uint32_t)std::max(1, std::min<int>(threads, (int)n_rows));
    std::atomic<uint32_t> next{i0};
    std::vector<std::thread> pool;
    auto work = [&]() { std::vector<double> dist; for (uint32_t i = next.fetch_add(1); i < i1; i = next.fetch_add(1)) do_row(i, dist); };
    for (uint32_t th = 1; th < nthreads; ++th) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    for (const std::string &f : row_fail) if (!f.empty()) return f;
    return "";
}

}  // namespace lashhost
