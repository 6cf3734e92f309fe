"""hyperminhash's expected_collisions(n, m) on the GPU (lash_hmh_pair_expected_collisions): the 65 536-cell sum that the
crate — and the host fallback — walk with four pow() per cell and pair, here one cell-probability vector per sketch and an
f64 MFMA matrix product per block.  Checked against the pure-Python restatement of the crate's loop (tests/pyref.py) and
against the library's own host path through lash_dist_rows."""
import time

import numpy as np
import pytest

import pyref as R

pytestmark = pytest.mark.gpu


def test_every_regime_matches_the_loop_and_the_closed_forms():
    import lash_amd
    rng = np.random.default_rng(4)
    ref = np.array([1.0, 2.0, 17.0, 1000.0, 49_999.5, 300_000.0, 524_288.0, 524_288.5, 3.0e6, 1.0e12, 2.0 ** 74, 2.0 ** 75])
    qry = np.concatenate([ref[::-1], rng.uniform(1, 524_288, 12), rng.uniform(6e5, 1e7, 3)])
    with lash_amd.Context(0) as ctx:
        got = ctx.hmh_pair_expected_collisions(ref, qry)
        again = ctx.hmh_pair_expected_collisions(ref[:5], qry)             # same queries: the cached vectors
        assert np.array_equal(again, got[:5])
        other = ctx.hmh_pair_expected_collisions(ref[:3], qry[:7])          # other queries: recomputed
        assert np.array_equal(other, got[:3, :7])
    assert got.shape == (len(ref), len(qry))
    checked_loop = 0
    for i, n in enumerate(ref):
        for j, m in enumerate(qry):
            big = max(n, m)
            if big > 2.0 ** 19 or (i + j) % 3 == 0:                        # the Python loop takes ~0.1 s per pair: a third of the small ones
                want = R.hmh_expected_collisions(float(n), float(m))
                checked_loop += big <= 2.0 ** 19
                assert got[i, j] == want or abs(got[i, j] - want) <= 1e-11 * max(abs(want), 1e-300), (n, m, got[i, j], want)
    assert checked_loop > 40


def test_distances_with_and_without_the_gpu_term_agree_and_the_gpu_is_the_fast_one():
    """lash_dist_rows computes the term itself when it is not handed in (the crate's loop, on the host).  Same distances;
    the point of the GPU path is time: 40 x 40 small sketches take the host minutes, the GPU milliseconds."""
    import lash_amd
    rng = np.random.default_rng(5)
    nr, nq = 6, 7
    rc, qc = rng.uniform(2e4, 5e5, nr), rng.uniform(2e4, 5e5, nq)
    n = rng.integers(6000, 16384, (nr, nq)).astype(np.uint32)
    with lash_amd.Context(0) as ctx:
        ec = ctx.hmh_pair_expected_collisions(rc, qc)
        c = (ec + rng.uniform(1, 300, (nr, nq))).astype(np.uint32)         # just above the term: the subtraction matters
        t0 = time.perf_counter()
        slow = lash_amd.dist_rows("hmh", 0, 16, 1, rc, qc, c_or_zero=c, n_counts=n)
        t_host = time.perf_counter() - t0
        fast = lash_amd.dist_rows("hmh", 0, 16, 1, rc, qc, c_or_zero=c, n_counts=n, hmh_ec=ec)
        assert np.all(slow < 1.0) and np.max(np.abs(slow - fast)) <= 1e-12
        # 1 000 x 1 000 small sketches: the whole block in well under a second on the GPU (the host needs > 4 ms per pair)
        big_r, big_q = rng.uniform(1e3, 5e5, 1000), rng.uniform(1e3, 5e5, 1000)
        ctx.hmh_pair_expected_collisions(big_r[:8], big_q[:8])              # warm-up (module load)
        t0 = time.perf_counter()
        full = ctx.hmh_pair_expected_collisions(big_r, big_q)
        t_gpu = time.perf_counter() - t0
        assert np.all(np.isfinite(full)) and np.all(full > 0)
        per_pair_host = t_host / (nr * nq)
        assert t_gpu < 5.0 and t_gpu / 1e6 < per_pair_host / 1000, (t_gpu, per_pair_host)
        print("host %.2f ms per pair; GPU %.3f s for 10^6 pairs" % (per_pair_host * 1e3, t_gpu))


def test_cli_dist_on_virus_sized_genomes(tmp_path):
    """`lash dist` on genomes far below 2^19 distinct k-mers (the regime where the crate walks 65 536 cells per pair):
    every row equals the pure-Python restatement, which does walk them."""
    import os
    import subprocess
    import host_lib as H
    import oracle_lib as O
    rng = np.random.default_rng(6)
    base = O.synth_genome(4000, 40_000)
    genomes = [base]
    for rate in (0.002, 0.02, 0.1):
        g = base.copy()
        idx = rng.random(len(g)) < rate
        g[idx] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(idx.sum()))
        genomes.append(g)
    genomes += [O.synth_genome(4001, 9_000), base[:15_000].copy(), O.synth_genome(4002, 120_000)]
    paths = []
    for i, g in enumerate(genomes):
        f = tmp_path / ("v%d.fa" % i)
        f.write_bytes(b">v\n" + g.tobytes() + b"\n")
        paths.append(str(f))
    (tmp_path / "l.txt").write_text("\n".join(paths) + "\n")
    r = subprocess.run([H.CLI, "sketch", "-f", "l.txt", "-o", "vir", "-k", "16"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([H.CLI, "dist", "-q", "vir", "-r", "vir", "-o", "d.tsv"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    imgs = [O.sketch_genomes(O.HMH, 16, 0, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0].tobytes() for g in genomes]
    assert all(R.hmh_cardinality(im) < 2.0 ** 19 for im in imgs)
    rows = (tmp_path / "d.tsv").read_text().strip().split("\n")[1:]
    assert len(rows) == len(genomes) * (len(genomes) + 1) // 2
    related = 0
    for ln in rows:
        a, b, d = ln.split("\t")
        i, j = paths.index(a), paths.index(b)
        want = R.mash_distance(R.hmh_similarity(imgs[j], imgs[i]), 16, 1, i == j)
        assert abs(float(d) - want) <= 1.1e-6, (i, j, d, want)
        related += 0 < float(d) < 1
    assert related >= 6
