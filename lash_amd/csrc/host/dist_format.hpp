// dist_format.hpp — the text `lash dist` prints for a block of reference rows (/root/reference/src/main.rs:429-471, print_dist):
// list form "Reference\tQuery\t{:.6}\n" per pair, or with --dm one matrix line "\n" + reference name + "\t{:.6}" per column.
// Shared by the C++ command line (dist.cpp) and, through liblash_host.so, by the multi-rank Python driver (lash_amd/allpairs.py),
// so that 5 * 10^9 rows (BASELINE configs[3]) are never formatted by an interpreter.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lashhost {

// "\t" + name + "\t" of every query column, made once per run (list form only)
std::vector<std::string> tabbed_names(const std::vector<std::string> &names);

// name -> small integer, the same integer for the same name on both sides: "q_name == r_name" (main.rs:452) as an int compare
void name_ids(const std::vector<std::string> &row_name, const std::vector<std::string> &col_name, std::vector<uint32_t> &row_id,
              std::vector<uint32_t> &col_id);

// Appends the text of ONE reference row: its first n_print columns, dist[c] the distance to column c, except where the column
// carries the row's name (col_id[c] == row_id), which prints 0 (main.rs:452-453).  qtab: tabbed_names() of the columns (list form).
void append_row(std::string &out, const std::string &rname, const std::vector<std::string> &qtab, uint32_t n_print, const double *dist,
                uint32_t row_id, const uint32_t *col_id, bool matrix);
// the same text into caller-provided memory of at least row_text_bound() bytes; returns the bytes written
size_t row_text_bound(const std::string &rname, const std::vector<std::string> &qtab, uint32_t n_print, bool matrix);
size_t format_row(char *dst, const std::string &rname, const std::vector<std::string> &qtab, uint32_t n_print, const double *dist,
                  uint32_t row_id, const uint32_t *col_id, bool matrix);

// One block of reference rows [i0, i1) from the GPU's pair statistics to text: per row the distances of its printed columns
// (lash_dist_rows: similarity -> Mash distance, main.rs:415-423) and append_row, on `threads` host threads (the reference's
// par_iter over reference sketches, utils.rs:150,248,342).  Tables are row-major [i1 - i0][ld]; row i prints columns
// [0, triangle ? min(i + 1, n_cols_total) : n_cols_total) (utils.rs:158-160).  row_name / row_card / row_id are indexed by the
// global row i, col_tab / col_name / col_card / col_id by column.  Returns "" or the message `lash dist` ends with; text row i - i0 = the row.
struct BlockTables {
    const uint32_t *c_or_zero = nullptr, *n_counts = nullptr;   // hmh: C, N; hll: zero registers of the union
    const double *sum_or_union = nullptr;                       // hll: sum of the union; ull: union estimate
    const double *hmh_ec = nullptr;                             // hmh: expected collisions of small pairs, or NULL
    uint64_t ld = 0;
};
// The text of a block lives in ONE buffer that its owner keeps from block to block (RowText): every row has a slot of its upper-bound
// size, rows are formatted into their slots in parallel and read back in order (data(r), size(r)).  A fresh std::string per row —
// megabytes each, hundreds per block — was mmap / page-fault / munmap churn that cost more than the formatting itself
// (single thread: 52 -> 21 ns per pair) and serialised the threads in the kernel.
struct RowText {
    std::vector<char> buf;                 // grows, never shrinks: its pages are touched once per run
    std::vector<size_t> off, len;          // per row of the last block
    size_t rows() const { return len.size(); }
    const char *data(size_t r) const { return buf.data() + off[r]; }
    size_t size(size_t r) const { return len[r]; }
};
std::string dist_block_rows(int algo, int p, int k, int model, bool fp32, const void *hll_bias, uint32_t i0, uint32_t i1, bool triangle,
                            uint32_t n_cols_total, const double *row_card, const double *col_card, const BlockTables &t,
                            const std::vector<std::string> &row_name, const std::vector<std::string> &col_name,
                            const std::vector<std::string> &col_tab, const uint32_t *row_id, const uint32_t *col_id, bool matrix, int threads,
                            RowText &text);

}  // namespace lashhost
