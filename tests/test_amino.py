"""Amino-acid sketching (/root/reference/src/utils.rs:43-55, 66-81, 511-563 — the `aa` branch, unreachable in
the reference because main.rs:198 hard-wires aa = false).  CPU: the oracle's restatement against an independent pure-Python
one (upper-casing, the RAW-length skip, filter_out_a, 5-bit codes, mask_aa_bits, which bytes each sketch type hashes).
GPU: the HIP path (lash_sketch_batch with LASH_F_AMINO, lash_sketch_files_raw, `lash sketch --aa`) against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import pyref as R

LETTERS = b"ACDEFGHIKLMNPQRSTVWY"
ALGO = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}


def aa_kmers(rec: bytes, k: int, zero_based=False):
    """masked k-mer values of one record, in order (utils.rs:519-552)"""
    s = rec.upper()                                               # to_ascii_uppercase (bytes.upper() is ASCII-only)
    if len(s) < k:
        return []
    codes = [LETTERS.index(bytes([c])) + (0 if zero_based else 1) for c in s if bytes([c]) in LETTERS]
    out = []
    for i in range(len(codes) - k + 1):
        v = 0
        for c in codes[i:i + k]:
            v = (v << 5) | c
        out.append(v & ((1 << (5 * k)) - 1))
    return out


def proteome(rng, n_rec, dirty=True):
    recs = []
    for _ in range(n_rec):
        n = int(rng.integers(0, 400))
        s = bytes(rng.choice(np.frombuffer(LETTERS, np.uint8), size=n))
        if dirty and n > 10:
            s = bytearray(s)
            for _ in range(int(rng.integers(0, 6))):
                s[int(rng.integers(0, n))] = int(rng.choice(np.frombuffer(b"BJOUXZ*-acdxy1 ", np.uint8)))
            s = bytes(s)
        recs.append(s)
    return recs


@pytest.mark.parametrize("algo,p", [("hmh", 0), ("hll", 10), ("ull", 9)])
def test_oracle_amino_equals_python_restatement(algo, p):
    rng = np.random.default_rng(31)
    recs = proteome(rng, 60) + [b"ACD", b"acdefghiklmnpqrstvwy" * 3, b"XXXXXXXXXXXXXXXXXXXXACDEF", b"", b"M"]
    seq = np.frombuffer(b"".join(recs), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    for k in (1, 3, 6, 7, 12):
        for zero in (False, True):
            lay = O.make_layout(aa_codes="zero" if zero else "one")
            got = O.sketch_genomes(ALGO[algo], k, p, 42, seq, off, np.array([0, len(recs)], np.uint64), layout=lay, amino=True)[0]
            vals = [v for r in recs for v in aa_kmers(r, k, zero)]
            want = R.sketch_from_masked_kmers(algo, p, 42, vals)
            assert bytes(got) == want, (algo, k, zero)
    # the raw-length rule: a record of k raw bytes of which fewer than k are residues is NOT skipped by length but yields nothing,
    # and a record whose raw length is below k is skipped even though ... it could not hold a k-mer anyway (same images)
    with pytest.raises(ValueError):
        O.sketch_genomes(O.HMH, 13, 0, 42, seq, off, np.array([0, len(recs)], np.uint64), amino=True)      # utils.rs:554 panic


@pytest.mark.gpu
@pytest.mark.parametrize("algo,p", [("hmh", 0), ("hll", 12), ("hll", 16), ("ull", 10), ("ull", 16), ("ull", 18), ("ull", 21)])   # (18, 21: binned tables)
def test_gpu_amino_equals_oracle(algo, p):
    import lash_amd
    rng = np.random.default_rng(37)
    genomes = [proteome(rng, 3000), proteome(rng, 1), [], proteome(rng, 9000, dirty=False), [b"MKV"], proteome(rng, 40)]
    seq, off, goff = lash_amd.records_to_arrays(genomes)
    with lash_amd.Context(0) as ctx:
        for k in (1, 5, 6, 7, 12):
            got = ctx.sketch_batch(algo, k, p, 42, seq, off, goff, flags=lash_amd.F_AMINO)
            want = O.sketch_genomes(ALGO[algo], k, p, 42, seq, off, goff, threads=4, amino=True)
            assert np.array_equal(got, want), (algo, p, k)
        # union into existing images; the other code table
        k = 4
        a = ctx.sketch_batch(algo, k, p, 42, seq, off, goff, flags=lash_amd.F_AMINO)
        half = len(genomes[0]) // 2
        g2 = [genomes[0][:half]] + genomes[1:]
        g3 = [genomes[0][half:]] + [[] for _ in genomes[1:]]
        s2, o2, f2 = lash_amd.records_to_arrays(g2)
        s3, o3, f3 = lash_amd.records_to_arrays(g3)
        b = ctx.sketch_batch(algo, k, p, 42, s2, o2, f2, flags=lash_amd.F_AMINO)
        b = ctx.sketch_batch(algo, k, p, 42, s3, o3, f3, flags=lash_amd.F_AMINO | lash_amd.F_ACCUMULATE, out=b)
        assert np.array_equal(a, b)
        ctx.set_layout("aa_codes=zero")
        got = ctx.sketch_batch(algo, 5, p, 42, seq, off, goff, flags=lash_amd.F_AMINO)
        want = O.sketch_genomes(ALGO[algo], 5, p, 42, seq, off, goff, threads=4, amino=True, layout=O.make_layout(aa_codes="zero"))
        assert np.array_equal(got, want)
        ctx.set_layout(None)
        with pytest.raises(lash_amd.LashError):
            ctx.sketch_batch(algo, 13, p, 42, seq, off, goff, flags=lash_amd.F_AMINO)


@pytest.mark.gpu
def test_cli_aa_flag(tmp_path):
    import host_lib as H
    rng = np.random.default_rng(41)
    files, paths = [], []
    for i in range(4):
        recs = proteome(rng, 500 + 300 * i)
        if i == 2:                                                # a protein FASTQ, why not
            data = b"".join(b"@p%d\n%s\n+\n%s\n" % (j, r, b"I" * len(r)) for j, r in enumerate(recs))
        else:
            data = b"".join(b">p%d some description\n%s\n" % (j, b"\n".join(r[x:x + 60] for x in range(0, len(r), 60)) if r else b"") for j, r in enumerate(recs))
        files.append(data)
        path = tmp_path / ("prot%d.%s" % (i, "fq" if i == 2 else "faa"))
        path.write_bytes(data)
        paths.append(str(path))
    (tmp_path / "l.txt").write_text("\n".join(paths) + "\n")
    for algo, k, p in (("hmh", 7, 10), ("ull", 5, 12)):
        out = str(tmp_path / ("aa_" + algo))
        r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "l.txt"), "-o", out, "-a", algo, "-k", str(k), "-p", str(p), "--aa", "-t", "2"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        blob = H.zstd_read(out + "_sketches.bin")
        want = O.sketch_files(ALGO[algo], k, p if algo != "hmh" else 0, 42, files, threads=2, amino=True)
        assert blob == want.tobytes(), algo
        import json
        assert json.load(open(out + "_parameters.json"))["molecule"] == "amino_acid"
    r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "l.txt"), "-o", "x", "-k", "13", "--aa"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 101 and "amino acid" in r.stderr
