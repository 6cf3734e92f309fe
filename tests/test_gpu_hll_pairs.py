"""HyperLogLog pair statistics, threshold-bitmap form (p >= 10; dist_kernels.hip: hll_pairs_bitmap_kernel) and byte-wise form
(p < 10, degenerate ranges): zero = #{max(a, b) == 0} and sum = sum 2^-max(a, b), the latter correctly rounded from the exact
rational — both forms must return the same bits (utils.rs:355-363: union + len())."""
import os
import subprocess
import sys
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _images(rng, n, p, kind):
    m = 1 << p
    if kind == "band":                                   # what sketches look like: a band of ~20 values, a few far above
        regs = rng.integers(6, 24, (n, m), dtype=np.uint8)
        hot = rng.random((n, m)) < 0.001
        regs[hot] = rng.integers(24, 64 - p + 2, int(hot.sum()), dtype=np.uint8)
    elif kind == "sparse":                               # small genomes: mostly empty registers
        regs = np.where(rng.random((n, m)) < 0.8, 0, rng.integers(1, 12, (n, m))).astype(np.uint8)
    elif kind == "full_range":
        regs = rng.integers(0, 64 - p + 2, (n, m), dtype=np.uint8)
    else:                                                # every register equal (range of one value)
        regs = np.full((n, m), 7, np.uint8)
    img = np.zeros((n, 33 + m), np.uint8)
    img[:, :33] = rng.integers(0, 256, (n, 33), dtype=np.uint8)      # header bytes are not looked at
    img[:, 33:] = regs
    return img


def _want(ref, qry):
    nr, nq = len(ref), len(qry)
    zero = np.zeros((nr, nq), np.uint32)
    s = np.zeros((nr, nq), np.float64)
    for i in range(nr):
        mx = np.maximum(ref[i, 33:][None, :], qry[:, 33:])
        zero[i] = (mx == 0).sum(axis=1)
        for j in range(nq):
            hist = np.bincount(mx[j], minlength=72)
            s[i, j] = float(sum(Fraction(int(c), 1 << r) for r, c in enumerate(hist) if c))
    return zero, s


@pytest.mark.parametrize("p", [8, 10, 12, 14, 16])
@pytest.mark.parametrize("kind", ["band", "sparse", "full_range", "flat"])
def test_pair_statistics_are_exact(p, kind):
    import lash_amd
    rng = np.random.default_rng(p * 10 + len(kind))
    nr, nq = (70, 130) if p <= 12 else (9, 67)
    ref, qry = _images(rng, nr, p, kind), _images(rng, nq, p, kind)
    if kind == "band":
        qry[3] = ref[2]                                  # identical sketches
        qry[5, 33:] = 0                                  # an empty one
    with lash_amd.Context(0) as ctx:
        z, s = ctx.hll_pair_union_stats(p, ref, qry)
        z2, s2 = ctx.hll_pair_union_stats(p, qry, qry)   # reference set == query set (one bitmap set)
    wz, ws = _want(ref, qry)
    assert np.array_equal(z, wz)
    assert np.array_equal(s.view(np.uint64), ws.view(np.uint64)), np.max(np.abs(s - ws))
    wz2, ws2 = _want(qry[:8], qry)
    assert np.array_equal(z2[:8], wz2) and np.array_equal(s2[:8].view(np.uint64), ws2.view(np.uint64))


def test_both_forms_return_the_same_bits(tmp_path):
    """the byte-wise kernel (LASH_HLL_PAIRS_BYTEWISE=1, read once per process) on the same inputs"""
    import lash_amd
    rng = np.random.default_rng(77)
    p = 14
    ref, qry = _images(rng, 40, p, "band"), _images(rng, 100, p, "band")
    np.save(tmp_path / "ref.npy", ref)
    np.save(tmp_path / "qry.npy", qry)
    with lash_amd.Context(0) as ctx:
        z, s = ctx.hll_pair_union_stats(p, ref, qry)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import lash_amd\n"
            "ref, qry = np.load(%r), np.load(%r)\n"
            "with lash_amd.Context(0) as ctx:\n"
            "    z, s = ctx.hll_pair_union_stats(14, ref, qry)\n"
            "np.save(%r, z); np.save(%r, s)\n" % (ROOT, str(tmp_path / "ref.npy"), str(tmp_path / "qry.npy"), str(tmp_path / "z.npy"), str(tmp_path / "s.npy")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, LASH_HLL_PAIRS_BYTEWISE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(tmp_path / "z.npy"), z)
    assert np.array_equal(np.load(tmp_path / "s.npy").view(np.uint64), s.view(np.uint64))
