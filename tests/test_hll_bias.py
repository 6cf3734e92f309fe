"""HLL++ bias-table regime of streaming_algorithms' len() (utils.rs:315, 355-363): the tables are external data
(include/lash_gfx950.h: lash_hll_bias), so these tests drive the mechanism with SYNTHETIC tables — file loader,
6-nearest-neighbour rule, refusal without tables — against tests/pyref.py.  Host arithmetic only: no GPU."""
import ctypes as C
import math

import numpy as np
import pytest

import lash_amd
import pyref as R


def _fake_tables(p, n=120, seed=0):
    """shaped like the published ones: raw estimates rising from ~0.7 m to ~5 m, bias falling from ~0.7 m to ~0"""
    rng = np.random.default_rng(seed + p)
    m = float(1 << p)
    raw = np.sort(rng.uniform(0.7 * m, 5.0 * m, n))
    bias = 0.7 * m * np.exp(-(raw - 0.7 * m) / m) + rng.normal(0, 0.01 * m, n)
    return raw.tolist(), bias.tolist()


def _write(path, tables, comment=True):
    with open(path, "w") as f:
        if comment:
            f.write("# synthetic tables for tests\n\n")
        for p, (raw, bias) in tables.items():
            f.write("p %d %d\n" % (p, len(raw)))
            for r, b in zip(raw, bias):
                f.write("%r %r\n" % (r, b))


def _regs(p, n_distinct, seed):
    """HLL registers of n_distinct random 64-bit hashes (low p bits = bucket, rank of the rest)"""
    rng = np.random.default_rng(seed)
    regs = np.zeros(1 << p, np.uint8)
    h = rng.integers(0, 2**63, n_distinct, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n_distinct, dtype=np.uint64)
    bucket = (h & np.uint64((1 << p) - 1)).astype(np.int64)
    rest = h >> np.uint64(p)
    rank = np.array([64 - p - int(x).bit_length() + 1 for x in rest], dtype=np.uint8)
    np.maximum.at(regs, bucket, rank)
    return regs


def test_bias_regime_with_and_without_tables(tmp_path):
    tables = {p: _fake_tables(p) for p in (10, 12, 14)}
    path = tmp_path / "bias.txt"
    _write(path, tables)
    tb = lash_amd.HllBias(path)
    assert tb.has(10) and tb.has(14) and not tb.has(11) and not tb.has(16)
    seen_regimes = set()
    for p in (10, 12, 14):
        m = 1 << p
        for frac in (0.05, 0.5, 0.9, 1.5, 3.0, 4.5, 6.0, 20.0):
            regs = _regs(p, int(frac * m), seed=int(frac * 100) + p)
            img = np.concatenate([np.zeros(33, np.uint8), regs])
            want = R.hll_len_from_regs(p, [int(x) for x in regs], tables)
            bare = R.hll_len_from_regs(p, [int(x) for x in regs], None)
            got = lash_amd.sketch_cardinality("hll", p, img, hll_bias=tb)
            assert got == want or abs(got - want) <= 1e-9 * abs(want), (p, frac)
            if bare is None:
                seen_regimes.add("bias")
                with pytest.raises(lash_amd.LashError) as e:
                    lash_amd.sketch_cardinality("hll", p, img)
                assert e.value.code == lash_amd.ERANGE
                with pytest.raises(lash_amd.LashError):
                    lash_amd.sketch_cardinality("hll", p, img, hll_bias=lash_amd.HllBias().set(11, *_fake_tables(11)))   # no table for THIS p
            else:
                seen_regimes.add("plain")
                assert lash_amd.sketch_cardinality("hll", p, img) == got
    assert seen_regimes == {"bias", "plain"}


def test_nearest_neighbours_ties_and_table_edges():
    """e below the first / above the last sample, and exactly between two samples (equal squared distances: the lower
    index wins, as a stable sort on the distance gives)."""
    p, m = 10, 1024.0
    raw = [1000.0 + 100.0 * i for i in range(20)]
    bias = [float((i * 7919) % 101) for i in range(20)]
    tables = {p: (raw, bias)}
    tb = lash_amd.HllBias().set(p, raw, bias)
    seen = set()
    for e in (900.0, 1300.0, 1800.0, 2300.0, 1949.999, 2650.0, 3000.0, 5000.0):   # 1300 / 1800 / 2300: the 6th and 7th nearest are equally far
        # registers are not needed to reach estimate_bias: use lash_dist_rows' hll branch with zero = 0, sum = alpha m^2 / e
        s = R.hll_alpha(p) * m * m / e
        ref_card = np.array([0.8 * e]); qry_card = np.array([0.7 * e])
        zero = np.array([0], np.uint32); ssum = np.array([s])
        e_back = R.hll_alpha(p) * m * m / s
        want_u = e_back - R.hll_estimate_bias(tables, p, e_back) if e_back <= 5 * m else e_back
        d = lash_amd.dist_rows("hll", p, 21, 1, ref_card, qry_card, c_or_zero=zero, sum_or_union=ssum, hll_bias=tb)
        sim = max((ref_card[0] + qry_card[0] - want_u) / want_u, 0.0)
        assert sim > 0
        frac = 2 * sim / (1 + sim)
        assert abs(float(d[0, 0]) - min(-math.log(frac) / 21, 1.0)) <= 1e-13, e
        seen.add(round(R.hll_estimate_bias(tables, p, e_back), 9))
    assert len(seen) >= 5                                   # the cases really pick different neighbour sets
    with pytest.raises(lash_amd.LashError):
        lash_amd.dist_rows("hll", p, 21, 1, np.array([1.0]), np.array([1.0]), c_or_zero=np.array([0], np.uint32),
                           sum_or_union=np.array([R.hll_alpha(p) * m * m / 2000.0]))


@pytest.mark.parametrize("text", ["p 10 6\n1 2\n3 4\n", "p 3 6\n" + "1 2\n" * 6, "1 2\n", "p 10 6\n" + "1 2\n" * 6 + "p 10 6\n" + "1 2\n" * 6,
                                  "p 10 6\n" + "1 x\n" * 6, "", "p 10 7\n" + "1 2\n" * 6])
def test_malformed_table_files_are_rejected(tmp_path, text):
    f = tmp_path / "bad.txt"
    f.write_text(text)
    with pytest.raises(lash_amd.LashError) as e:
        lash_amd.HllBias(f)
    assert e.value.code == lash_amd._lib.EFORMAT
    with pytest.raises(lash_amd.LashError) as e:
        lash_amd.HllBias(tmp_path / "absent.txt")
    assert e.value.code == lash_amd.EINVAL
