// ull_estimators.h — UltraLogLog distinct-count estimators as functions of the register HISTOGRAM (256 bins), shared by the
// gfx950 pair kernel (dist_kernels.hip) and the C++ host side.  Replaces, for `lash dist -a ull`,
//     ull.get_distinct_count_estimate()            (FGRA)      /root/reference/src/utils.rs:214, 266
//     MaximumLikelihoodEstimator.estimate(&ull)    (ML)        /root/reference/src/utils.rs:215, 267
// of crate ultraloglog 0.1.6 (Cargo.lock:2012), a port of hash4j's UltraLogLog.  [PARITY UNPINNED: the crate is not in the
// reference tree.]  Restated from O. Ertl, "UltraLogLog: A Practical and More Space-Efficient Alternative to HyperLogLog for
// Approximate Distinct Counting" (VLDB 2024) and the published hash4j implementation of its estimators:
//
//   register r > 0:  u = (r >> 2) - p + 2 is the largest update value seen (P(update = k) = 2^-k), bit 1 / bit 0 of r say
//   whether u-1 / u-2 were seen too.  Values below 1 do not exist, which makes r in {0, 4p-4, 4p, 4p+2} special ("small
//   range"); u = 65-p is saturated ("large range", r >= 252; unreachable below ~2^50 distinct elements).
//
//   FGRA  n = lambda_p * (sum_i g(r_i))^(-1/tau),  g(r) = eta_(r&3) * 2^(-tau*u),  lambda_p = m^(1+1/tau) / (1 + v(1+tau)/(2m));
//         special registers contribute the conditional expectation of g given z = e^(-n/m), z itself the ML estimate from
//         the counts of the four small-range values (quadratic in z^(1/4)); the derivations are spelled out at each function.
//   ML    maximises  e^(-x a) * prod_j (1 - e^(-x / 2^j))^(b_j)  with x = n/(2m); a and b_j are sums over the registers
//         (exact integers), solved with Ertl's bracketed secant iteration, then divided by (1 + c/m).
//
// Everything is a pure function of (histogram, p): the GPU builds one histogram per (reference, query) pair of the merged
// registers pack(unpack(a) | unpack(b)) (UltraLogLog::merge, utils.rs:261) and evaluates the estimator in the same kernel.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LASH_HD __host__ __device__ __forceinline__
#else
#define LASH_HD inline
#endif

namespace lash {
namespace ull {

// constants of the FGRA estimator (paper, Table of optimal coefficients for the 2-bit register extension)
constexpr double ETA_0 = 4.663135422063788;
constexpr double ETA_1 = 2.1378502137958524;
constexpr double ETA_2 = 2.781144650979996;
constexpr double ETA_3 = 0.9824082545153715;
constexpr double TAU = 0.8194911375910897;
constexpr double V = 0.6118931496978437;
constexpr double ETA_X = ETA_0 - ETA_1 - ETA_2 + ETA_3;
// ML: first-order bias correction and the solver's tolerance scale (1 / sqrt of the Fisher information per register)
constexpr double ML_BIAS_CORRECTION_CONSTANT = 0.48147376527720065;
constexpr double INV_SQRT_FISHER_INFORMATION = 0.7608621002725182;

LASH_HD double eta(uint32_t low2) { return low2 == 0 ? ETA_0 : low2 == 1 ? ETA_1 : low2 == 2 ? ETA_2 : ETA_3; }

// E[eta | the two bits below the leading one are set with probabilities 1 - y and 1 - y^2]   (y = "not seen" probability of
// the nearer one; the farther one has twice the rate):   eta_X * ((y + e23)(y^2 + e13) + e3012)
//   = y*y^2*eta0 + y(1-y^2)*eta1 + (1-y)y^2*eta2 + (1-y)(1-y^2)*eta3
LASH_HD double psi(double y) { const double y2 = y * y; return y * y2 * ETA_X + y2 * (ETA_2 - ETA_3) + y * (ETA_1 - ETA_3) + ETA_3; }

// contribution of an EMPTY register given z = e^(-n/m): the largest (virtual) update value is -j with probability
// z^(2^j - 1) * (1 - z^(2^j)), weight 2^(tau*j), and the two bits below it follow psi(z^(2^(j+1)))
LASH_HD double sigma(double z)
{
    if (z <= 0.0) return ETA_3;
    if (z >= 1.0) return INFINITY;
    double pow_z = z, next = z * z, s = 0.0, pow_tau = 1.0;
    const double two_tau = pow(2.0, TAU);
    for (;;) {
        const double old = s;
        s += pow_tau * (pow_z - next) * psi(next);
        if (!(s > old)) return s / z;
        pow_z = next;
        next = next * next;
        pow_tau *= two_tau;
    }
}

// z = e^(-n/m) from the counts of empty registers (c0) and of those whose largest update value is 1 (c4), 2 without 1 (c8),
// 2 with 1 (c10): with x = z^(1/4) their probabilities are x^4, x^2(1-x^2), x^3(1-x), x(1-x^2)(1-x), all others 1-x;
// the likelihood equation is  alpha x^2 + beta x - gamma = 0.
LASH_HD double small_range_z(double c0, double c4, double c8, double c10, double m)
{
    const double alpha = m + 3.0 * (c0 + c4 + c8 + c10), beta = m - c0 - c4, gamma = 4.0 * c0 + 2.0 * c4 + 3.0 * c8 + c10;
    const double x = (sqrt(beta * beta + 4.0 * alpha * gamma) - beta) / (2.0 * alpha);
    const double x2 = x * x;
    return x2 * x2;
}

// saturated registers (largest representable update value 65 - p; r = 252 + low bits), as published in hash4j
// [restated from memory of the published code; unreachable for inputs below ~2^50 distinct elements]
LASH_HD double large_range_z(double w0, double w1, double w2, double w3, double m)
{
    const double alpha = m + 3.0 * (w0 + w1 + w2 + w3), beta = w0 + w1 + 2.0 * (w2 + w3), gamma = m + 2.0 * w0 + w2 - w3;
    return sqrt((sqrt(beta * beta + 4.0 * alpha * gamma) - beta) / (2.0 * alpha));
}
LASH_HD double phi(double z, double z_square)
{
    const double pow2_mtau = pow(2.0, -TAU);
    if (z <= 0.0) return 0.0;
    if (z >= 1.0) return ETA_0 / (pow(2.0, TAU) * (2.0 * pow(2.0, TAU) - 1.0));
    double prev = z_square, pz = z, next = sqrt(pz);
    double pr = ETA_X * (pow(4.0, -TAU) / (2.0 - pow2_mtau)) / (1.0 + next);
    double ps = psi(pz) / 1.0;
    (void)prev;
    double s = next * (ps + ps) * pr;
    for (;;) {
        prev = pz;
        pz = next;
        const double old = s;
        next = sqrt(pz);
        const double nps = psi(pz);
        pr *= pow2_mtau / (1.0 + next);
        s += next * ((nps + nps) - (pz + next) * ps) * pr;
        if (!(s > old)) return s;
        ps = nps;
    }
}

// hist(r): how many of the 2^p registers hold value r (0..255)
template <class Hist>
LASH_HD double fgra(const Hist &hist, int p)
{
    const double m = (double)(1u << p);
    const uint32_t off = 4u * (uint32_t)p + 4u;                 // first regular register value: largest update value 3
    double sum = 0.0;
    for (uint32_t r = off; r < 252u; ++r) {
        const uint32_t c = hist(r);
        if (c) sum += (double)c * eta(r & 3u) * pow(2.0, -TAU * (double)((r >> 2) - (uint32_t)p + 2u));
    }
    double c0 = 0.0, c4 = 0.0, c8 = 0.0, c10 = 0.0;
    for (uint32_t r = 0; r < off && r < 256u; ++r) {
        const double c = (double)hist(r);
        if (c == 0.0) continue;
        if (r + 8u < off) c0 += c;                              // r == 0 (nothing else occurs below 4p-4)
        else if (r + 8u == off) c4 += c;                        // 4p-4: largest update value 1
        else if (r + 4u == off) c8 += c;                        // 4p:   2, without 1
        else if (r + 2u == off) c10 += c;                       // 4p+2: 2, with 1
    }
    if (c0 > 0.0 || c4 > 0.0 || c8 > 0.0 || c10 > 0.0) {
        const double z = small_range_z(c0, c4, c8, c10, m);
        const double q2 = pow(2.0, -TAU), q4 = pow(4.0, -TAU);
        if (c0 > 0.0) sum += c0 * sigma(z);
        if (c4 > 0.0) sum += c4 * q2 * psi(z);                                         // bits below: update values 0, -1
        if (c8 > 0.0) sum += c8 * q4 * (z * (ETA_0 - ETA_1) + ETA_1);                  // bit 1 clear, bit 0 = virtual value 0
        if (c10 > 0.0) sum += c10 * q4 * (z * (ETA_2 - ETA_3) + ETA_3);                // bit 1 set
    }
    if (p <= 62) {
        const double w0 = (double)hist(252u), w1 = (double)hist(253u), w2 = (double)hist(254u), w3 = (double)hist(255u);
        if (w0 > 0.0 || w1 > 0.0 || w2 > 0.0 || w3 > 0.0) {
            const double z = large_range_z(w0, w1, w2, w3, m), root_z = sqrt(z);
            const double q = pow(2.0, -TAU);
            double s = phi(root_z, z) * (w0 + w1 + w2 + w3);
            s += z * (1.0 + root_z) * (w0 * ETA_0 + w1 * ETA_1 + w2 * ETA_2 + w3 * ETA_3);
            s += root_z * ((w0 + w1) * (z * q * (ETA_0 - ETA_2) + q * ETA_2) + (w2 + w3) * (z * q * (ETA_1 - ETA_3) + q * ETA_3));
            sum += s * pow(q, (double)(65 - p)) / ((1.0 + root_z) * (1.0 + z));
        }
    }
    const double factor = pow(m, 1.0 + 1.0 / TAU) / (1.0 + V * (1.0 + TAU) / (2.0 * m));
    return factor * pow(sum, -1.0 / TAU);
}

// Ertl's solver for  a = sum_j b[j] / 2^j / (e^(x / 2^j) - 1)   ("New cardinality estimation algorithms for HyperLogLog
// sketches", alg. 8, as used by hash4j's DistinctCountUtil): secant steps on g(x) = sum_j b[j] h(x / 2^j) + a x with
// h(x) = x / (e^x - 1) evaluated by the doubling recurrence h(2x) = (x + h(x)(1 - h(x))) / (x + 1 - h(x)).
LASH_HD double solve_ml(double a, const uint32_t *b, int n, double rel_err)
{
    if (a == 0.0) return INFINITY;
    int k_max = n;
    while (k_max >= 0 && b[k_max] == 0u) --k_max;
    if (k_max < 0) return 0.0;
    int k_min = k_max;
    double s1 = (double)b[k_max], s2 = ldexp((double)b[k_max], k_max);
    for (int k = k_max - 1; k >= 0; --k) {
        if (b[k]) { s1 += (double)b[k]; s2 += ldexp((double)b[k], k); k_min = k; }
    }
    double g_prev = 0.0, x;
    if (s2 <= 1.5 * a) x = s1 / (0.5 * s2 + a);
    else x = log1p(s2 / a) * (s1 / s2);
    double dx = x;
    while (dx > x * rel_err) {
        int kappa;
        (void)frexp(x, &kappa);                                  // x = f * 2^kappa, f in [0.5, 1)  ->  getExponent(x) = kappa - 1
        kappa += 1;                                              // getExponent(x) + 2
        const int top = k_max > kappa ? k_max : kappa;
        double xp = ldexp(x, -(top + 1));
        const double xp2 = xp * xp;
        double h = xp - xp2 / 3.0 + (xp2 * xp2) * (1.0 / 45.0 - xp2 / 472.5);
        for (int k = kappa - 1; k >= k_max; --k) {
            const double hp = 1.0 - h;
            h = (xp + h * hp) / (xp + hp);
            xp += xp;
        }
        double g = (double)b[k_max] * h;
        for (int k = k_max - 1; k >= k_min; --k) {
            const double hp = 1.0 - h;
            h = (xp + h * hp) / (xp + hp);
            xp += xp;
            g += (double)b[k] * h;
        }
        g += x * a;
        if (g_prev < g && g <= s1) dx *= (g - s1) / (g_prev - g);
        else dx = 0.0;
        x += dx;
        g_prev = g;
    }
    return x;
}

// register r -> its share of `a` (in units of 2^-64 * 2m, kept as an exact 64-bit integer like hash4j) and of b[]:
//   r > 4p+2:  a += (7 - 4 y0 - 2 y1) * 2^(1-u),  b[u-1] += 1, b[u-2] += y1, b[u-3] += y0     (y1 = bit 1, y0 = bit 0 of r)
template <class Hist>
LASH_HD double ml(const Hist &hist, int p)
{
    const double m = (double)(1u << p);
    uint32_t b[64];
    for (int i = 0; i < 64; ++i) b[i] = 0u;
    uint64_t sum = 0;
    const int off = 4 * p + 4;
    for (int r = 0; r < 256; ++r) {
        const uint32_t c = hist((uint32_t)r);
        if (!c) continue;
        const int r2 = r - off;
        uint64_t ret;
        if (r2 < 0) {
            uint64_t t = 4;
            if (r2 == -2 || r2 == -8) { b[0] += c; t -= 2; }
            if (r2 == -2 || r2 == -4) { b[1] += c; t -= 1; }
            ret = t << (62 - p);
        } else {
            const int k = r2 >> 2;
            const uint64_t y0 = (uint64_t)(r & 1), y1 = (uint64_t)((r >> 1) & 1);
            uint64_t t = 0xE000000000000000ull;
            t -= y0 << 63;
            t -= y1 << 62;
            b[k] += (uint32_t)y0 * c;
            b[k + 1] += (uint32_t)y1 * c;
            b[k + 2] += c;
            ret = t >> (k + p);
        }
        sum += ret * (uint64_t)c;                                // < 2^64: m registers of at most 2^(64-p) each
    }
    if (sum == 0) return hist(0u) == (1u << p) ? 0.0 : INFINITY;
    b[63 - p] += b[64 - p];                                      // the saturated value has the probability of its predecessor
    const double factor = 2.0 * m;
    const double a = (double)sum * factor * 5.421010862427522e-20;    // 2^-64
    return factor * solve_ml(a, b, 63 - p, 0.001 * INV_SQRT_FISHER_INFORMATION / sqrt(m)) / (1.0 + ML_BIAS_CORRECTION_CONSTANT / m);
}

}  // namespace ull
}  // namespace lash
