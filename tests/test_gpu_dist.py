"""GPU tests of the dist side: the HyperMinHash pair-statistics kernel against numpy, and the
`lash dist` command line against a pure-Python restatement of hyperminhash's similarity + the Mash distance
(main.rs:415-423).  Row order is nondeterministic in the reference, so rows are compared as a set."""
import os
import subprocess

import numpy as np
import pytest

import host_lib as H
import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hmh_pair_counts_match_numpy():
    import lash_amd
    ctx = lash_amd.Context(0)
    rng = np.random.default_rng(9)
    nr, nq = 37, 21
    ref = rng.integers(0, 5, size=(nr, 16384), dtype=np.uint16)        # small alphabet -> many equal / zero registers
    qry = rng.integers(0, 5, size=(nq, 16384), dtype=np.uint16)
    qry[3] = ref[5]
    qry[4] = 0
    c, n = ctx.hmh_pair_counts(ref.view(np.uint8).reshape(nr, -1), qry.view(np.uint8).reshape(nq, -1))
    want_c = ((ref[:, None, :] == qry[None, :, :]) & (ref[:, None, :] != 0)).sum(axis=2)
    want_n = ((ref[:, None, :] != 0) | (qry[None, :, :] != 0)).sum(axis=2)
    assert np.array_equal(c, want_c) and np.array_equal(n, want_n)
    ctx.close()


def _mutated(seq: np.ndarray, rate: float, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    out = seq.copy()
    idx = rng.random(len(seq)) < rate
    out[idx] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(idx.sum()))
    return out


@pytest.mark.parametrize("matrix,model,fp32", [(False, 1, False), (True, 1, False), (False, 0, True)])
def test_lash_dist_cli_hmh(tmp_path, matrix, model, fp32):
    # five related genomes (mutation rates 0 .. 10 %) so that distances span the interesting range
    base = O.synth_genome(77, 400_000)
    genomes = [base, _mutated(base, 0.001, 1), _mutated(base, 0.01, 2), _mutated(base, 0.1, 3), O.synth_genome(78, 400_000)]
    paths = []
    for i, g in enumerate(genomes):
        p = tmp_path / ("g%d.fa" % i)
        p.write_bytes(b">g\n" + g.tobytes() + b"\n")
        paths.append(str(p))
    (tmp_path / "all.txt").write_text("\n".join(paths) + "\n")
    (tmp_path / "two.txt").write_text("\n".join(paths[:2]) + "\n")
    env = dict(os.environ)
    for pre, lst in (("refs", "all.txt"), ("qry", "two.txt")):
        r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / lst), "-o", pre, "-k", "16"], cwd=tmp_path, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
    imgs = [O.sketch_genomes(O.HMH, 16, 0, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0].tobytes() for g in genomes]

    def expected(ri, qi):
        return R.mash_distance(R.hmh_similarity(imgs[qi], imgs[ri]), 16, model, paths[ri] == paths[qi])

    flags = (["--dm"] if matrix else []) + (["--fp32"] if fp32 else []) + ["-m", str(model)]
    # all-vs-all on the same files: lower triangle incl. the diagonal, every unordered pair exactly once
    r = subprocess.run([H.CLI, "dist", "-q", "refs", "-r", "refs", "-o", "d_self.txt"] + flags, cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "Distances computed." in r.stdout
    text = (tmp_path / "d_self.txt").read_text()
    # the same in blocks of 2 reference rows per GPU call (how 10^5 x 10^5 runs stay bounded): identical file
    r = subprocess.run([H.CLI, "dist", "-q", "refs", "-r", "refs", "-o", "d_blk.txt", "--block-rows", "2"] + flags, cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "d_blk.txt").read_text() == text
    tol = 2e-6 if fp32 else 1.1e-6
    if not matrix:
        lines = text.strip().split("\n")
        assert lines[0] == "Reference\tQuery\tDistance"
        rows = {}
        for ln in lines[1:]:
            a, b, d = ln.split("\t")
            rows[frozenset((a, b))] = float(d)
        assert len(rows) == 5 * 6 // 2 and len(lines) - 1 == 15
        # rows, and which name of a pair is the Reference, in the reference's map order (name_order.hpp)
        order = R.hashbrown_name_order(paths)
        assert [tuple(ln.split("\t")[:2]) for ln in lines[1:]] == [(paths[order[a]], paths[order[b]]) for a in range(5) for b in range(a + 1)]
        for i in range(5):
            for j in range(i + 1):
                assert abs(rows[frozenset((paths[i], paths[j]))] - expected(i, j)) <= tol, (i, j)
        d01 = rows[frozenset((paths[0], paths[1]))]
        d03 = rows[frozenset((paths[0], paths[3]))]
        d04 = rows[frozenset((paths[0], paths[4]))]
        assert 0 < d01 < rows[frozenset((paths[0], paths[2]))] < d03 <= d04 <= 1.0
    else:
        order = R.hashbrown_name_order(paths)
        lines = text.split("\n")
        assert lines[0] == "".join("\t" + paths[j] for j in order)
        assert len(lines) == 6 and not text.endswith("\n")
        for a, ln in enumerate(lines[1:]):
            cells = ln.split("\t")
            assert cells[0] == paths[order[a]] and len(cells) == a + 2
            for b, d in enumerate(cells[1:]):
                assert abs(float(d) - expected(order[a], order[b])) <= tol
        # --file-order: the same numbers with rows and columns as the list file has them
        r = subprocess.run([H.CLI, "dist", "-q", "refs", "-r", "refs", "-o", "d_fo.txt", "--file-order"] + flags, cwd=tmp_path, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        fo = (tmp_path / "d_fo.txt").read_text().split("\n")
        assert fo[0] == "".join("\t" + p for p in paths)
        for i, ln in enumerate(fo[1:]):
            cells = ln.split("\t")
            assert cells[0] == paths[i] and len(cells) == i + 2
            for j, d in enumerate(cells[1:]):
                assert abs(float(d) - expected(i, j)) <= tol
    # query set vs reference set (different files): full rectangle
    r = subprocess.run([H.CLI, "dist", "-q", "qry", "-r", "refs", "-o", "d_qr.txt", "-m", str(model)], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    lines = (tmp_path / "d_qr.txt").read_text().strip().split("\n")
    assert len(lines) == 1 + 5 * 2
    ro, qo = R.hashbrown_name_order(paths), R.hashbrown_name_order(paths[:2])
    assert [tuple(ln.split("\t")[:2]) for ln in lines[1:]] == [(paths[i], paths[j]) for i in ro for j in qo]
    for ln in lines[1:]:
        a, b, d = ln.split("\t")
        assert abs(float(d) - R.mash_distance(R.hmh_similarity(imgs[paths.index(b)], imgs[paths.index(a)]), 16, model, a == b)) <= 1.1e-6
    # mismatched k is refused like the reference's panic (main.rs:368-370)
    r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "two.txt"), "-o", "k21", "-k", "21"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0
    r = subprocess.run([H.CLI, "dist", "-q", "k21", "-r", "refs"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "same k" in r.stderr


def test_hll_pair_union_stats_match_numpy():
    import lash_amd
    ctx = lash_amd.Context(0)
    rng = np.random.default_rng(4)
    for p, nr, nq in ((4, 3, 2), (10, 19, 33), (14, 17, 5)):
        m = 1 << p
        ref = np.zeros((nr, 33 + m), np.uint8)
        qry = np.zeros((nq, 33 + m), np.uint8)
        ref[:, :33] = rng.integers(0, 256, size=(nr, 33))                   # headers must be ignored
        ref[:, 33:] = rng.integers(0, 40, size=(nr, m)) * (rng.random((nr, m)) < 0.7)
        qry[:, 33:] = rng.integers(0, 62 - p, size=(nq, m)) * (rng.random((nq, m)) < 0.5)
        qry[0, 33:] = 0
        zero, usum = ctx.hll_pair_union_stats(p, ref, qry)
        u = np.maximum(ref[:, None, 33:], qry[None, :, 33:]).astype(np.int64)
        assert np.array_equal(zero, (u == 0).sum(axis=2))
        want = np.array([[float(np.sum(np.ldexp(1.0, -u[i, j]).astype(np.longdouble))) for j in range(nq)] for i in range(nr)])
        assert np.array_equal(usum, want)                                    # exact sum, rounded once
    ctx.close()


@pytest.mark.parametrize("matrix,model,p", [(False, 1, 10), (True, 0, 12)])
def test_lash_dist_cli_hll(tmp_path, matrix, model, p):
    base = O.synth_genome(177, 300_000)
    genomes = [base, _mutated(base, 0.002, 1), _mutated(base, 0.02, 2), O.synth_genome(178, 300_000)]
    paths = []
    for i, g in enumerate(genomes):
        f = tmp_path / ("h%d.fa" % i)
        f.write_bytes(b">g\n" + g.tobytes() + b"\n")
        paths.append(str(f))
    (tmp_path / "all.txt").write_text("\n".join(paths) + "\n")
    env = dict(os.environ)
    r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "all.txt"), "-o", "hl", "-k", "21", "-a", "hll", "-p", str(p)],
                       cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    imgs = [O.sketch_genomes(O.HLL, 21, p, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0].tobytes() for g in genomes]
    flags = (["--dm"] if matrix else []) + ["-m", str(model)]
    r = subprocess.run([H.CLI, "dist", "-q", "hl", "-r", "hl", "-o", "d.txt"] + flags, cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    text = (tmp_path / "d.txt").read_text()

    def expected(i, j):
        return R.mash_distance(R.hll_similarity(p, imgs[i], imgs[j]), 21, model, i == j)

    got = {}
    if not matrix:
        for ln in text.strip().split("\n")[1:]:
            a, b, d = ln.split("\t")
            got[frozenset((paths.index(a), paths.index(b)))] = float(d)
    else:
        order = R.hashbrown_name_order(paths)
        assert text.split("\n")[0] == "".join("\t" + paths[j] for j in order)
        for a, ln in enumerate(text.split("\n")[1:]):
            cells = ln.split("\t")
            assert cells[0] == paths[order[a]] and len(cells) == a + 2
            for b, d in enumerate(cells[1:]):
                got[frozenset((order[a], order[b]))] = float(d)
    assert len(got) == 4 * 5 // 2
    for i in range(4):
        for j in range(i + 1):
            assert abs(got[frozenset((i, j))] - expected(i, j)) <= 1.1e-6, (i, j)
    assert 0 < got[frozenset((0, 1))] < got[frozenset((0, 2))] < got[frozenset((0, 3))] <= 1.0


def test_lash_dist_hll_refuses_the_bias_table_regime(tmp_path):
    """A 20 kbp genome at p = 14: the raw estimate is below 5 * 2^14, where streaming_algorithms subtracts a bias read from
    the HLL++ tables.  Those tables are not available here, so the command must fail loudly, never estimate differently."""
    g = O.synth_genome(3, 20_000)
    (tmp_path / "s.fa").write_bytes(b">s\n" + g.tobytes() + b"\n")
    (tmp_path / "l.txt").write_text(str(tmp_path / "s.fa") + "\n")
    env = dict(os.environ)
    for algo, pre in (("hll", "sm"),):
        r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "l.txt"), "-o", pre, "-a", algo, "-p", "14"], cwd=tmp_path, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
    env.pop("LASH_HLL_BIAS", None)
    r = subprocess.run([H.CLI, "dist", "-q", "sm", "-r", "sm"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "bias tables" in r.stderr


@pytest.mark.parametrize("via", ["flag", "env"])
def test_lash_dist_hll_small_range_with_tables(tmp_path, via):
    """The same regime WITH tables (--hll-bias / $LASH_HLL_BIAS): 6-nearest-neighbour bias subtracted from the raw estimate,
    per sketch and per union (utils.rs:315, 355-363).  The tables here are synthetic — the real ones are external data
    (tools/ref_probe/extract_hll_bias.py) — so this checks the mechanism against tests/pyref.py, not the numbers of HLL++."""
    p, k = 14, 21
    m = float(1 << p)
    rng = np.random.default_rng(8)
    raw = np.sort(rng.uniform(0.7 * m, 5.0 * m, 200))
    bias = 0.6 * m * np.exp(-(raw - 0.7 * m) / m)
    tables = {p: (raw.tolist(), bias.tolist())}
    with open(tmp_path / "bias.txt", "w") as f:
        f.write("# synthetic\np %d %d\n" % (p, len(raw)) + "".join("%r %r\n" % (float(a), float(b)) for a, b in zip(raw, bias)))
    base = O.synth_genome(31, 40_000)                              # ~40 k distinct 21-mers: between the linear-counting threshold and 5m
    genomes = [base, _mutated(base, 0.01, 1), _mutated(base, 0.05, 2), O.synth_genome(32, 30_000)]
    paths = []
    for i, g in enumerate(genomes):
        f = tmp_path / ("s%d.fa" % i)
        f.write_bytes(b">s\n" + g.tobytes() + b"\n")
        paths.append(str(f))
    (tmp_path / "l.txt").write_text("\n".join(paths) + "\n")
    env = dict(os.environ)
    env.pop("LASH_HLL_BIAS", None)
    r = subprocess.run([H.CLI, "sketch", "-f", "l.txt", "-o", "sm", "-a", "hll", "-p", str(p), "-k", str(k)], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    imgs = [O.sketch_genomes(O.HLL, k, p, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0].tobytes() for g in genomes]
    assert all(R.hll_len_from_regs(p, im[33:]) is None for im in imgs)            # every sketch is in the refused regime without tables
    cmd = [H.CLI, "dist", "-q", "sm", "-r", "sm", "-o", "d.txt"]
    if via == "flag":
        cmd += ["--hll-bias", "bias.txt"]
    else:
        env["LASH_HLL_BIAS"] = str(tmp_path / "bias.txt")
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    rows = (tmp_path / "d.txt").read_text().strip().split("\n")[1:]
    assert len(rows) == 4 * 5 // 2
    for ln in rows:
        a, b, d = ln.split("\t")
        i, j = paths.index(a), paths.index(b)
        want = R.mash_distance(R.hll_similarity(p, imgs[i], imgs[j], tables), k, 1, i == j)
        assert abs(float(d) - want) <= 1.1e-6, (i, j)
    # a malformed table file is an error, not a silent fallback
    (tmp_path / "broken.txt").write_text("p 14 200\n1 2\n")
    r = subprocess.run([H.CLI, "dist", "-q", "sm", "-r", "sm", "--hll-bias", "broken.txt"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "bias tables" in r.stderr


def _ull_regs(img, p):
    return np.frombuffer(img, np.uint8)[8:]


@pytest.mark.parametrize("p", [5, 12, 15, 16, 18])
def test_ull_pair_union_estimates_match_host_estimators(p):
    """GPU: histogram of pack(unpack(a) | unpack(b)) per pair + FGRA / ML in the kernel (narrow form p <= 15, wide form above)
    == the host entry on the oracle-merged sketch (the same ull_estimators.h compiled for the host)."""
    import lash_amd
    ctx = lash_amd.Context(0)
    n_gen = 5 if p < 16 else 3
    base = O.synth_genome(277, 250_000)
    genomes = [base, _mutated(base, 0.003, 1), _mutated(base, 0.05, 2), O.synth_genome(278, 60_000), np.frombuffer(b"ACGTTGCATGCATCGATCGGATTACA", np.uint8)][:n_gen]
    imgs = np.stack([O.sketch_genomes(O.ULL, 16, p, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0] for g in genomes])
    empty = imgs[:1].copy()
    empty[0, 8:] = 0
    ref = np.concatenate([imgs, empty])
    for est, tol in (("fgra", 1e-9), ("ml", 2e-4)):
        got = ctx.ull_pair_union_estimates(p, ref, imgs, estimator=est)
        assert got.shape == (len(ref), len(imgs))
        for i in range(len(ref)):
            for j in range(len(imgs)):
                merged = O.merge_images(O.ULL, p, ref[i], imgs[j])
                want = lash_amd.ull_estimate(merged[8:], p, est)
                assert got[i, j] == pytest.approx(want, rel=tol, abs=1e-9), (p, est, i, j)
        # a sketch merged with itself is itself
        for j in range(len(imgs)):
            assert got[j, j] == pytest.approx(lash_amd.ull_estimate(imgs[j][8:], p, est), rel=tol)
    ctx.close()


@pytest.mark.parametrize("matrix,model,p,est", [(False, 1, 12, "fgra"), (True, 0, 10, "ml"), (False, 1, 14, "ml")])
def test_lash_dist_cli_ull(tmp_path, matrix, model, p, est):
    """`lash dist` on UltraLogLog sketches (utils.rs:186-288) vs the per-register Python restatement of merge + estimator +
    inclusion-exclusion + Mash distance.  ML: the product stops its iteration at a relative step of 7.6e-4 / sqrt(m)."""
    base = O.synth_genome(377, 300_000)
    genomes = [base, _mutated(base, 0.002, 1), _mutated(base, 0.02, 2), O.synth_genome(378, 300_000)]
    paths = []
    for i, g in enumerate(genomes):
        f = tmp_path / ("u%d.fa" % i)
        f.write_bytes(b">g\n" + g.tobytes() + b"\n")
        paths.append(str(f))
    (tmp_path / "all.txt").write_text("\n".join(paths) + "\n")
    env = dict(os.environ)
    r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "all.txt"), "-o", "ul", "-k", "16", "-a", "ull", "-p", str(p)],
                       cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    imgs = [O.sketch_genomes(O.ULL, 16, p, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0].tobytes() for g in genomes]
    estimate = R.ull_fgra if est == "fgra" else R.ull_ml
    card = [estimate(list(im[8:]), p) for im in imgs]
    flags = (["--dm"] if matrix else []) + ["-m", str(model), "-e", est]
    r = subprocess.run([H.CLI, "dist", "-q", "ul", "-r", "ul", "-o", "d.txt", "--devices", "0,0"] + flags, cwd=tmp_path, capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, r.stderr
    text = (tmp_path / "d.txt").read_text()

    def expected(i, j):
        u = estimate(list(R.ull_merge(imgs[i][8:], imgs[j][8:])), p)
        return R.mash_distance((card[i] + card[j] - u) / u, 16, model, i == j)

    got = {}
    if not matrix:
        for ln in text.strip().split("\n")[1:]:
            a, b, d = ln.split("\t")
            got[frozenset((paths.index(a), paths.index(b)))] = float(d)
    else:
        order = R.hashbrown_name_order(paths)
        assert text.split("\n")[0] == "".join("\t" + paths[j] for j in order)
        for a, ln in enumerate(text.split("\n")[1:]):
            cells = ln.split("\t")
            assert cells[0] == paths[order[a]] and len(cells) == a + 2
            for b, d in enumerate(cells[1:]):
                got[frozenset((order[a], order[b]))] = float(d)
    assert len(got) == 4 * 5 // 2
    tol = 1.1e-6                                                 # both estimators: only the 6-decimal print rounds
    for i in range(4):
        for j in range(i + 1):
            assert abs(got[frozenset((i, j))] - expected(i, j)) <= tol, (i, j, got[frozenset((i, j))], expected(i, j))
    assert 0 < got[frozenset((0, 1))] < got[frozenset((0, 2))] < got[frozenset((0, 3))] <= 1.0
    # a bad estimator name is refused like the reference's panic (utils.rs:216)
    r = subprocess.run([H.CLI, "dist", "-q", "ul", "-r", "ul", "-e", "median"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "fgra or ml" in r.stderr


@pytest.mark.parametrize("algo,p", [("hmh", 0), ("ull", 10)])
def test_lash_dist_with_repeated_names(tmp_path, algo, p):
    """A list file that names a genome several times gives a names JSON with repeats; the reference's maps keep ONE entry per name
    (the last sketch, utils.rs:111-127).  `lash dist` used to size its cardinality tables by the map and index them by file
    position: a 400-entry list with 3 distinct names overran the heap.  Same-files and reference-vs-query, map order
    and --file-order."""
    gs = [O.synth_genome(90 + i, 120_000) for i in range(3)]
    paths = []
    for i, g in enumerate(gs):
        q = tmp_path / ("r%d.fa" % i)
        q.write_bytes(b">g\n" + g.tobytes() + b"\n")
        paths.append(str(q))
    many = [paths[i % 3] for i in range(400)]
    (tmp_path / "many.txt").write_text("\n".join(many) + "\n")
    (tmp_path / "three.txt").write_text("\n".join(paths) + "\n")
    (tmp_path / "dup.txt").write_text("\n".join([paths[0], paths[0], paths[1]]) + "\n")
    env = dict(os.environ)
    pflag = ["-p", str(p)] if algo != "hmh" else []
    for pre, lst in (("many", "many.txt"), ("three", "three.txt"), ("dup", "dup.txt")):
        r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / lst), "-o", pre, "-k", "16", "-a", algo] + pflag, cwd=tmp_path, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr

    def run(q, r_, *extra):
        out = "d_%s_%s%s.txt" % (q, r_, "_fo" if extra else "")
        r = subprocess.run([H.CLI, "dist", "-q", q, "-r", r_, "-o", out, "-t", "3"] + list(extra), cwd=tmp_path, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return [tuple(ln.split("\t")) for ln in (tmp_path / out).read_text().strip().split("\n")[1:]]
    base = {frozenset((a, b)): d for a, b, d in run("three", "three")}          # the three genomes without repeats
    assert len(base) == 6
    # 400 names, 3 distinct: the triangle of a 3-entry map
    rows = run("many", "many")
    assert len(rows) == 6 and {frozenset((a, b)): d for a, b, d in rows} == base
    order = R.hashbrown_name_order(many)
    assert len(order) == 3
    assert [(a, b) for a, b, _ in rows] == [(many[order[i]], many[order[j]]) for i in range(3) for j in range(i + 1)]
    # reference-vs-query with repeats on both sides: 3 x 2 map entries
    rows = run("dup", "many")
    assert len(rows) == 6 and all(base[frozenset((a, b))] == d for a, b, d in rows)
    rows = run("many", "dup")
    assert len(rows) == 6 and all(base[frozenset((a, b))] == d for a, b, d in rows)
    # --file-order keeps every listed entry: 3 rows (triangle 1 + 2 + 3), the repeated name against itself prints 0
    rows = run("dup", "dup", "--file-order")
    assert [(a, b) for a, b, _ in rows] == [(paths[0], paths[0]), (paths[0], paths[0]), (paths[0], paths[0]), (paths[1], paths[0]), (paths[1], paths[0]),
                                            (paths[1], paths[1])]
    assert [d for _, _, d in rows[:3]] == ["0.000000"] * 3 and rows[3][2] == rows[4][2] == base[frozenset((paths[0], paths[1]))]
