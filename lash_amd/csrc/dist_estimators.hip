// dist_estimators.hip — the O(sketches) and O(pairs) arithmetic of `lash dist`, host only (no GPU needed), behind the C ABI
// so that the C++ command line (host/dist.cpp) and the multi-rank Python driver (lash_amd/allpairs.py) run the same code.
//
// Reference: /root/reference/src/utils.rs:150-180 (hmh), 248-282 (ull), 342-369 (hll); main.rs:415-423 (distance).
// Every estimator is restated from the published algorithm its crate ports [PARITY UNPINNED; tools/ref_probe]:
//   hyperminhash 0.1.4      LogLog-beta cardinality + expected-collision correction of axiomhq/hyperminhash
//   streaming_algorithms    HLL++ len(): linear counting below the per-precision threshold, else alpha m^2 / sum, minus
//                           (estimate <= 5m) the mean bias of the 6 nearest raw estimates in the HLL++ empirical tables.
//                           The tables are Monte-Carlo output that is not in this image or this repository: they come
//                           from a file (lash_hll_bias_load; tools/ref_probe/extract_hll_bias.py writes it from the
//                           crate's source), and WITHOUT them that regime is REFUSED (LASH_ERANGE), never approximated
//   ultraloglog 0.1.6       FGRA / ML (ull_estimators.h)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/lash_gfx950.h"
#include "lash_common.h"

namespace {

constexpr int HP = 14, HQ = 6, HR = 10;
constexpr uint32_t HM = 1u << HP;

double hmh_beta(double ez)
{
    const double zl = std::log(ez + 1.0);
    return -0.370393911 * ez + 0.070471823 * zl + 0.17393686 * std::pow(zl, 2) + 0.16339839 * std::pow(zl, 3) +
           -0.09237745 * std::pow(zl, 4) + 0.03738027 * std::pow(zl, 5) + -0.005384159 * std::pow(zl, 6) +
           0.00042419 * std::pow(zl, 7);
}

double hmh_card_from_sums(double ez, double sum)
{
    const double m = (double)HM;
    const double alpha = 0.7213 / (1.0 + 1.079 / m);
    return alpha * m * (m - ez) / (hmh_beta(ez) + sum);
}

double hmh_cell_sum(double n, double m)
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
    return x;
}

double hmh_approx_expected_collisions(double n, double m)
{
    double out;
    if (lash::hmh_ec_closed_form(n, m, &out)) return out;
    return lash::hmh_ec_cell_walk(n, m);
}

// streaming_algorithms 0.3.3 len() thresholds (HLL++, Heule et al.), p = 4..18
constexpr double HLL_THRESHOLD[15] = {10, 20, 40, 80, 220, 400, 900, 1800, 3100, 6500, 11500, 20000, 50000, 120000, 350000};

double hll_alpha(int p)
{
    switch (p) {
    case 4: return 0.673;
    case 5: return 0.697;
    case 6: return 0.709;
    default: return 0.7213 / (1.0 + 1.079 / (double)(1u << p));
    }
}

}  // namespace

namespace lash {

bool hmh_ec_closed_form(double n, double m, double *out)
{
    // (constants: 2^(2^q + r) = 2^74, 2^(p + 5) = 2^19, 2^(p - r) = 16; the crate writes them as powf calls)
    if (n < m) std::swap(n, m);
    if (n > 18889465931478580854784.0) { *out = 1.8446744073709552e19; return true; }                 // u64::MAX
    if (n > 524288.0) {
        const double t = (1.0 + n) / m;
        const double d = (4.0 * n / m) / (t * t);
        *out = 0.169919487159739093975315012348 * 16.0 * d + 0.5;
        return true;
    }
    return false;
}

double hmh_ec_from_cell_sum(double x) { return (x * (double)HP + 0.5) / (double)HP; }

// hyperminhash's cardinality() from the histogram of the registers' 6-bit leading-zero fields (lash_sketch_set_cardinalities:
// the histogram comes from the GPU).  The crate adds 2^-lz register by register; while no register has lz > 39 every partial
// sum of that loop is exact (multiples of 2^-39 below 2^14 fit 53 bits), so any order gives the same bits.  Otherwise
// *exact = false and the caller falls back to the register-order loop (lash_hmh_cardinality).
double hmh_cardinality_from_hist(const uint32_t *hist64, bool *exact)
{
    double sum = 0.0;
    bool ok = true;
    for (int lz = 0; lz < 64; ++lz) {
        if (!hist64[lz]) continue;
        if (lz > 39) ok = false;
        sum += (double)hist64[lz] * std::ldexp(1.0, -lz);
    }
    if (exact) *exact = ok;
    return hmh_card_from_sums((double)hist64[0], sum);
}

double hmh_ec_cell_walk(double n, double m)
{
    if (n < m) std::swap(n, m);
    return hmh_ec_from_cell_sum(hmh_cell_sum(n, m));
}

}  // namespace lash

// p = 4..18: raw estimate -> bias samples, as the HLL++ appendix publishes them (rawEstimateData / biasData)
struct lash_hll_bias {
    std::vector<double> raw[15], bias[15];
};

namespace {

// estimate_bias: mean bias of the 6 samples whose raw estimate is nearest to e (squared distance, ties to the lower
// index, summed nearest first)
bool hll_estimate_bias(const lash_hll_bias *t, int p, double e, double &out)
{
    if (!t || t->raw[p - 4].size() < 6) return false;
    const std::vector<double> &raw = t->raw[p - 4], &bias = t->bias[p - 4];
    std::pair<double, size_t> best[7];
    size_t n = 0;
    for (size_t i = 0; i < raw.size(); ++i) {
        const double d = (e - raw[i]) * (e - raw[i]);
        size_t at = n;
        while (at > 0 && d < best[at - 1].first) { best[at] = best[at - 1]; --at; }   // strict <: an equal distance stays behind the lower index
        if (at < 6) best[at] = {d, i};
        if (n < 6) ++n;
    }
    double s = 0.0;
    for (size_t j = 0; j < 6; ++j) s += bias[best[j].second];
    out = s / 6.0;
    return true;
}

bool hll_len(int p, uint64_t zero, double sum, const lash_hll_bias *tables, double &out)
{
    const double m = (double)(1u << p);
    if (zero > 0) {
        const double h = m * std::log(m / (double)zero);
        if (h <= HLL_THRESHOLD[p - 4]) { out = h; return true; }
    }
    const double e = hll_alpha(p) * m * m / sum;
    if (e <= 5.0 * m) {                                              // bias-corrected regime
        double b;
        if (!hll_estimate_bias(tables, p, e, b)) return false;       // tables absent: refuse
        out = e - b;
        return true;
    }
    out = e;
    return true;
}

template <class T>
T compute_distance(T frac, int k, int model)
{
    const T kk = (T)k;
    // frac == 0 (no similarity left after the collision correction: nearly every pair of an all-vs-all): both models give exactly 1 —
    // -ln(0) / k = +inf -> min(.., 1) = 1;  1 - 0^(1/k) = 1 — without the libm call (glibc's pow(0, y) alone is 45 ns)
    if (frac == (T)0) return (T)1;
    if (model == 1) { const T d = -std::log(frac) / kk; return d < (T)1 ? d : (T)1; }      // (-frac.ln() / k).min(1)
    return (T)1 - std::pow(frac, (T)1 / kk);
}

}  // namespace

namespace lash {

int hll_cardinality_from_hist(const uint32_t *hist256, int p, const lash_hll_bias *tables, double *out)
{
    double sum = 0.0;
    for (int r = 255; r >= 0; --r)
        if (hist256[r]) sum += (double)hist256[r] * std::ldexp(1.0, -r);      // exact powers of two, largest exponent first
    return hll_len(p, hist256[0], sum, tables, *out) ? LASH_OK : LASH_ERANGE;
}

}  // namespace lash

extern "C" {

double lash_hmh_cardinality(const uint8_t *regs, int big_endian)
{
    if (!regs) return -1.0;
    double sum = 0.0, ez = 0.0;
    for (uint32_t i = 0; i < HM; ++i) {
        const uint32_t reg = big_endian ? (regs[2 * i + 1] | (regs[2 * i] << 8)) : (regs[2 * i] | (regs[2 * i + 1] << 8));
        const uint32_t lz = reg >> (16 - HQ);
        if (lz == 0) ez += 1.0;
        sum += std::ldexp(1.0, -(int)lz);                       // == 1 / 2^lz exactly
    }
    return hmh_card_from_sums(ez, sum);
}

// Text format: '#' comment lines; "p <p> <n>" then n lines "<raw estimate> <bias>", for any subset of p = 4..18.
int lash_hll_bias_load(const char *path, lash_hll_bias **out)
{
    if (!path || !out) return LASH_EINVAL;
    *out = nullptr;
    FILE *f = fopen(path, "r");
    if (!f) return LASH_EINVAL;
    lash_hll_bias *t = new lash_hll_bias();
    char line[512];
    int cur = -1;
    long left = 0;
    bool ok = true;
    while (ok && fgets(line, sizeof line, f)) {
        const char *s = line;
        while (*s == ' ' || *s == '\t') ++s;
        if (*s == '#' || *s == '\n' || *s == '\r' || *s == 0) continue;
        if (*s == 'p') {
            int p = 0; long n = 0;
            ok = left == 0 && sscanf(s + 1, "%d %ld", &p, &n) == 2 && p >= 4 && p <= 18 && n >= 6 && n < (1l << 20) && t->raw[p - 4].empty();
            cur = p; left = n;
        } else {
            double r, b;
            ok = cur >= 4 && left > 0 && sscanf(s, "%lf %lf", &r, &b) == 2;
            if (ok) { t->raw[cur - 4].push_back(r); t->bias[cur - 4].push_back(b); --left; }
        }
    }
    fclose(f);
    if (!ok || left != 0 || cur < 0) { delete t; return LASH_EFORMAT; }
    *out = t;
    return LASH_OK;
}

int lash_hll_bias_from_arrays(lash_hll_bias **inout, int p, const double *raw, const double *bias, uint32_t n)
{
    if (!inout || !raw || !bias || p < 4 || p > 18 || n < 6) return LASH_EINVAL;
    if (!*inout) *inout = new lash_hll_bias();
    (*inout)->raw[p - 4].assign(raw, raw + n);
    (*inout)->bias[p - 4].assign(bias, bias + n);
    return LASH_OK;
}

void lash_hll_bias_free(lash_hll_bias *t) { delete t; }

int lash_hll_bias_has(const lash_hll_bias *t, int p) { return t && p >= 4 && p <= 18 && t->raw[p - 4].size() >= 6 ? 1 : 0; }

int lash_hll_cardinality(const uint8_t *regs, int p, const lash_hll_bias *tables, double *out)
{
    if (!regs || !out || p < 4 || p > 16) return LASH_EINVAL;
    uint32_t hist[256] = {0};
    for (size_t i = 0, m = (size_t)1 << p; i < m; ++i) hist[regs[i]]++;
    return lash::hll_cardinality_from_hist(hist, p, tables, out);
}

int lash_dist_rows(int algo, int p, int k, int model, int fp32, uint32_t n_ref, uint32_t n_qry, const double *ref_card,
                   const double *qry_card, const uint32_t *c_or_zero, const uint32_t *n_counts, const double *sum_or_union,
                   const lash_hll_bias *tables, const double *hmh_ec, double *out_dist, uint64_t *bad_pair)
{
    if (k < 1 || k > 32 || (model != 0 && model != 1) || !ref_card || !qry_card || !out_dist) return LASH_EINVAL;
    if (algo == LASH_HMH ? (!c_or_zero || !n_counts) : algo == LASH_HLL ? (!c_or_zero || !sum_or_union || p < 4 || p > 16)
                         : algo == LASH_ULL ? !sum_or_union : true)
        return LASH_EINVAL;
    for (uint32_t i = 0; i < n_ref; ++i) {
        for (uint32_t j = 0; j < n_qry; ++j) {
            const uint64_t at = (uint64_t)i * n_qry + j;
            double sim = 0.0;
            if (algo == LASH_HLL) {                                                               // utils.rs:352-365
                double u;
                if (!hll_len(p, c_or_zero[at], sum_or_union[at], tables, u)) { if (bad_pair) *bad_pair = at; return LASH_ERANGE; }
                sim = (ref_card[i] + qry_card[j] - u) / u;
            } else if (algo == LASH_ULL) {                                                        // utils.rs:256-274
                const double u = sum_or_union[at];
                sim = (ref_card[i] + qry_card[j] - u) / u;
            } else {
                const double c = (double)c_or_zero[at], n = (double)n_counts[at];
                if (c != 0.0) {                                                                   // Sketch::similarity
                    double ec;                                            // O(1) unless both sketches are small: then the caller's, if given
                    if (!lash::hmh_ec_closed_form(qry_card[j], ref_card[i], &ec)) ec = hmh_ec ? hmh_ec[at] : lash::hmh_ec_cell_walk(qry_card[j], ref_card[i]);
                    sim = c < ec ? 0.0 : (c - ec) / n;
                }
            }
            // hmh / hll: `.max(0.0)` (utils.rs:164, 362) — f64::max drops a NaN; ull: `if similarity < 0.0 {0.0} else {similarity}`
            // (utils.rs:272-273) keeps it: two empty sketches give 0/0, model 1 then prints 1 (f64::min drops the NaN), model 0 NaN
            if (algo == LASH_ULL) sim = sim < 0.0 ? 0.0 : sim;
            else if (!(sim >= 0.0)) sim = 0.0;
            const double frac = 2.0 * sim / (1.0 + sim);                                          // utils.rs:165-167
            out_dist[at] = fp32 ? (double)compute_distance<float>((float)frac, k, model) : compute_distance<double>(frac, k, model);
        }
    }
    return LASH_OK;
}

}  // extern "C"
