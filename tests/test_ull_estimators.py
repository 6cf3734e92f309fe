"""UltraLogLog distinct-count estimators of the dist side (utils.rs:186-288; ultraloglog 0.1.6 = hash4j port) — CPU tests of
the product's host entry `lash_ull_estimate` (the same ull_estimators.h the gfx950 pair kernel compiles):
  * against the per-register Python restatement (tests/pyref.py) on simulated sketches, tight tolerance;
  * statistically: over many simulated sketches the estimators are unbiased and their spread is the published one
    (FGRA: sqrt(0.6119 / m), ML: 0.7609 / sqrt(m)), from a handful of elements to thousands per register — the check that
    does not depend on anybody's memory of the constants.  [The crate itself is not in the reference tree: PARITY UNPINNED.]"""
import math

import numpy as np
import pytest

import lash_amd
import pyref as R


def simulate(rng, n, p):
    """ULL registers after n distinct uniformly hashed elements: idx uniform, update value geometric, state = pack(OR of bits)"""
    m = 1 << p
    idx = rng.integers(0, m, size=n)
    k = np.minimum(rng.geometric(0.5, size=n), 65 - p)         # P(k) = 2^-k, saturating at 65 - p
    bits = np.zeros(m, dtype=np.uint64)
    np.bitwise_or.at(bits, idx, np.uint64(1) << (k + p - 2).astype(np.uint64))
    regs = np.zeros(m, dtype=np.uint8)
    for i in np.nonzero(bits)[0]:
        regs[i] = R.ull_pack(int(bits[i]))
    return regs


@pytest.mark.parametrize("p", [3, 6, 10])
def test_host_estimators_match_per_register_restatement(p):
    rng = np.random.default_rng(100 + p)
    m = 1 << p
    for n in [0, 1, 2, 7, m // 2, m, 3 * m, 40 * m, 5000 * m]:
        regs = simulate(rng, n, p)
        f, want = lash_amd.ull_estimate(regs, p, "fgra"), R.ull_fgra([int(x) for x in regs], p)
        assert f == pytest.approx(want, rel=1e-10, abs=1e-12), (p, n)
        ml, want = lash_amd.ull_estimate(regs, p, "ml"), R.ull_ml([int(x) for x in regs], p)
        # the product stops its secant iteration at a relative step of 7.6e-4 / sqrt(m); bisection runs to the end
        assert ml == pytest.approx(want, rel=2e-3 / math.sqrt(m), abs=1e-12), (p, n)


@pytest.mark.parametrize("est,sigma", [("fgra", math.sqrt(0.6118931496978437)), ("ml", 0.7608621002725182)])
def test_estimators_are_unbiased_with_the_published_spread(est, sigma):
    p, trials = 8, 400
    m = 1 << p
    rng = np.random.default_rng(7)
    for n in [3, 40, m, 4 * m, 50 * m, 2000 * m]:
        rel = np.array([lash_amd.ull_estimate(simulate(rng, n, p), p, est) / n - 1.0 for _ in range(trials)])
        sd_theory = sigma / math.sqrt(m)
        if n >= 4 * m:                                          # the asymptotic regime the published error refers to
            assert abs(rel.mean()) < 4 * sd_theory / math.sqrt(trials) + 2e-3, (est, n, rel.mean())
            assert 0.8 * sd_theory < rel.std() < 1.25 * sd_theory, (est, n, rel.std(), sd_theory)
        else:                                                   # small range: still (nearly) unbiased, never worse than the asymptotic spread by much
            assert abs(rel.mean()) < 4 * max(rel.std(), 1e-9) / math.sqrt(trials) + 5e-3, (est, n, rel.mean())
            assert rel.std() < 1.6 * sd_theory + 1e-9, (est, n, rel.std())


def test_merge_then_estimate_is_inclusion_exclusion_consistent():
    """|A u B| from the merged sketch vs the truth, on sets with a known overlap"""
    p, rng = 10, np.random.default_rng(3)
    m = 1 << p

    def sketch(keys):
        bits = np.zeros(m, dtype=np.uint64)
        h = (keys * np.uint64(0x9E3779B97F4A7C15)) ^ (keys >> np.uint64(29))
        h = (h * np.uint64(0xBF58476D1CE4E5B9)) ^ (h >> np.uint64(32))
        for hv in h:
            hv = int(hv)
            idx = hv >> (64 - p)
            t = (~((~hv & (2**64 - 1)) << p)) & (2**64 - 1)
            nlz = 64 - t.bit_length()
            bits[idx] |= np.uint64(1 << (nlz + p - 1))
        return bytes(R.ull_pack(int(b)) if b else 0 for b in bits)
    keys = rng.integers(1, 2**62, size=60_000).astype(np.uint64)
    a, b = sketch(keys[:40_000]), sketch(keys[20_000:])
    u = R.ull_merge(a, b)
    assert u == sketch(keys)                                    # merge of sketches == sketch of the union
    for est in ("fgra", "ml"):
        got = lash_amd.ull_estimate(np.frombuffer(u, np.uint8), p, est)
        assert abs(got / 60_000 - 1) < 0.08
