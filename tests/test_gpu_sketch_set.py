"""GPU tests of the resident dist side (include/lash_gfx950.h: lash_sketch_set_*; SURVEY §8(f) row f2 at BASELINE configs[3]
scale): HyperMinHash pair counts through register bit planes against numpy and against the u16-pair kernel, per-member
cardinalities from GPU histograms against the per-image host entries, row blocks / triangle / member order for all three
sketch types against the whole-matrix entries."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _numpy_counts(ref, qry):
    c = np.zeros((len(ref), len(qry)), np.uint32)
    n = np.zeros_like(c)
    for i in range(len(ref)):
        c[i] = ((ref[i][None, :] == qry) & (ref[i][None, :] != 0)).sum(axis=1)
        n[i] = ((ref[i][None, :] != 0) | (qry != 0)).sum(axis=1)
    return c, n


@pytest.mark.parametrize("full", [False, True])
def test_hmh_planes_match_numpy(full):
    """several row workgroups (256 rows each), several column tiles, ragged edges; registers from a small alphabet so that
    equal / zero registers are frequent; `full`: no zero register anywhere (the 17-instruction form, N = 16384)"""
    import lash_amd
    rng = np.random.default_rng(11 + full)
    nr, nq = 300, 150
    lo = 1 if full else 0
    ref = rng.integers(lo, 6, size=(nr, 16384), dtype=np.uint16)
    qry = rng.integers(lo, 6, size=(nq, 16384), dtype=np.uint16)
    ref[7] = rng.integers(1, 65535, size=16384, dtype=np.uint16)       # all 16 planes in use
    qry[9] = ref[7]
    qry[10] = ref[7]
    qry[10, ::3] ^= 0x8000
    if not full:
        qry[4] = 0
        ref[5] = 0
    with lash_amd.Context(0) as ctx:
        c, n = ctx.hmh_pair_counts(ref.view(np.uint8).reshape(nr, -1), qry.view(np.uint8).reshape(nq, -1))
    wc, wn = _numpy_counts(ref, qry)
    assert np.array_equal(c, wc) and np.array_equal(n, wn)
    if full:
        assert (n == 16384).all()


def test_planes_and_word_kernels_agree():
    """the same call through the bit-plane kernel (default) and the u16-pair kernel (LASH_HMH_PAIRS_WORDS=1, read once per process)"""
    code = ("import numpy as np, lash_amd, sys\n"
            "rng = np.random.default_rng(5)\n"
            "a = rng.integers(0, 4, size=(70, 16384), dtype=np.uint16); b = rng.integers(0, 4, size=(33, 16384), dtype=np.uint16)\n"
            "ctx = lash_amd.Context(0)\n"
            "c, n = ctx.hmh_pair_counts(a.view(np.uint8).reshape(70, -1), b.view(np.uint8).reshape(33, -1))\n"
            "s = ctx.sketch_set('hmh', 0, a.view(np.uint8).reshape(70, -1)); s.prepare()\n"
            "t = s.pair_block(3, 64, n_cols=64, triangle=True)\n"
            "sys.stdout.buffer.write(c.tobytes() + n.tobytes() + np.tril(t['c_or_zero'], 3).tobytes())\n")
    outs = []
    for knob in ("", "1"):
        env = dict(os.environ, PYTHONPATH=ROOT)
        if knob:
            env["LASH_HMH_PAIRS_WORDS"] = knob
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and len(outs[0]) > 70 * 33 * 8


def _images(algo, k, p, n, L=30_000, seed=42):
    """n real sketches (oracle-made): genomes of different sizes, some related"""
    a = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}[algo]
    out = []
    base = O.synth_genome(500, L)
    for g in range(n):
        if g % 3 == 0:
            seq = base[: L - 97 * g].copy()
            seq[g::11] = ord("A")
        else:
            seq = O.synth_genome(600 + g, L // (1 + g % 4))
        out.append(O.sketch_genomes(a, k, p, seed, seq, np.array([0, len(seq)], np.uint64), np.array([0, 1], np.uint64))[0])
    return np.stack(out)


@pytest.mark.parametrize("algo,p", [("hmh", 0), ("hll", 12), ("hll", 8), ("ull", 11)])
def test_set_cardinalities_equal_the_per_image_entries(algo, p):
    import lash_amd
    imgs = _images(algo, 16, p, 9)
    order = np.array([4, 0, 8, 8, 2], np.uint32)
    bias = None
    if algo == "hll":
        bias = lash_amd.HllBias()
        for pp in (8, 12):
            raw = np.linspace(0.5, 6.0, 40) * (1 << pp)
            bias.set(pp, raw, 0.01 * raw)
    with lash_amd.Context(0) as ctx:
        for est in (("fgra", "ml") if algo == "ull" else ("fgra",)):
            s = ctx.sketch_set(algo, p, imgs, order)
            got = s.cardinalities(est, bias)
            want = np.array([lash_amd.sketch_cardinality(algo, p, imgs[i], None, est, bias) for i in order])
            assert np.array_equal(got, want), (algo, est, got, want)
            s.free()


@pytest.mark.parametrize("algo,p,est", [("hmh", 0, "fgra"), ("hll", 12, "fgra"), ("hll", 8, "fgra"), ("ull", 11, "fgra"), ("ull", 11, "ml")])
def test_row_blocks_triangle_and_order_equal_the_whole_matrix_entries(algo, p, est):
    import lash_amd
    n = 21
    imgs = _images(algo, 16, p, n)
    rng = np.random.default_rng(3)
    order = rng.permutation(n).astype(np.uint32)
    with lash_amd.Context(0) as ctx:
        if algo == "hmh":
            c, m = ctx.hmh_pair_counts(imgs[order], imgs[order])
            whole = dict(c_or_zero=c, n_counts=m)
        elif algo == "hll":
            z, s_ = ctx.hll_pair_union_stats(p, imgs[order], imgs[order])
            whole = dict(c_or_zero=z, sum_or_union=s_)
        else:
            whole = dict(sum_or_union=ctx.ull_pair_union_estimates(p, imgs[order], imgs[order], est))
        s = ctx.sketch_set(algo, p, imgs, order)
        for prepared in (False, True):
            if prepared:
                s.prepare()
            for r0, r1 in ((0, 5), (5, 6), (6, n)):
                tri = s.pair_block(r0, r1, n_cols=r1, triangle=True, estimator=est)
                rect = s.pair_block(r0, r1, estimator=est)
                for key, w in whole.items():
                    assert np.array_equal(rect[key], w[r0:r1]), (algo, key, prepared)
                    for i in range(r0, r1):                        # the triangle: columns <= row
                        assert np.array_equal(tri[key][i - r0, : i + 1], w[i, : i + 1]), (algo, key, prepared, i)
        # two different sets (reference x query), the query one adopted from device memory
        import torch
        q_dev = torch.from_numpy(np.ascontiguousarray(imgs[:7])).cuda()
        q = ctx.sketch_set(algo, p, q_dev)
        s.prepare(q)
        got = s.pair_block(2, 19, qry=q, estimator=est)
        if algo == "hmh":
            c, m = ctx.hmh_pair_counts(imgs[order][2:19], imgs[:7])
            assert np.array_equal(got["c_or_zero"], c) and np.array_equal(got["n_counts"], m)
        elif algo == "hll":
            z, s_ = ctx.hll_pair_union_stats(p, imgs[order][2:19], imgs[:7])
            assert np.array_equal(got["c_or_zero"], z) and np.array_equal(got["sum_or_union"], s_)
        else:
            assert np.array_equal(got["sum_or_union"], ctx.ull_pair_union_estimates(p, imgs[order][2:19], imgs[:7], est))
        q.free()
        s.free()
