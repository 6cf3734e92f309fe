"""CPU tests of the oracle: pin it against the XXH3 golden vectors, SURVEY Appendix B's worked
example, the naive pure-Python restatement (tests/pyref.py) and the committed fixture digests."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import pyref as R
from fastx import read_fastx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_xxh3_golden_vectors():
    g = json.load(open(os.path.join(GOLD, "xxh3_vectors.json")))
    n = 0
    for si, seed in enumerate(int(s) for s in g["seeds"]):
        for v, h in g["xxh3_64_of_le8"][si]:
            assert O.xxh3_64_8b(int(v), seed) == int(h)
            assert R.xxh3_64_8b(int(v), seed) == int(h)
            n += 1
        for v, h in g["xxh3_128_of_le4"][si]:
            lo, hi = O.xxh3_128_4b(int(v), seed)
            assert (hi << 64) | lo == int(h)
            lo, hi = R.xxh3_128_4b(int(v), seed)
            assert (hi << 64) | lo == int(h)
            n += 1
    assert n == 6 * 512


def test_a_rank_39_kmer_met_by_the_fuzzer():
    """tests/fuzz_gpu.py, seed 44 iteration 187 (round 4): a synthetic genome held the 31-mer 0x13399b149020b766, whose xxh3_64 under
    seed 42 is 0x0000000002025fbf (python-xxhash, libxxhash 0.8.2, in the build container) — 38 leading zeros, one k-mer in 2^38.  At
    p = 16 its register gets rank 39 > 53 - p: the HyperLogLog `sum` corner (tests/test_gpu_hll_corner.py) met in the wild."""
    v, seed = 0x13399b149020b766, 42
    assert O.xxh3_64_8b(v, seed) == 0x0000000002025fbf == R.xxh3_64_8b(v, seed)
    try:
        import xxhash
        assert xxhash.xxh3_64_intdigest(v.to_bytes(8, "little"), seed=seed) == 0x0000000002025fbf
    except ImportError:
        pass
    km = "".join("ACGT"[(v >> (2 * (30 - i))) & 3] for i in range(31)).encode()
    assert O.record_kmers(km, 31)[0] == v                         # (its own canonical form)
    img = O.sketch_genomes(O.HLL, 31, 16, seed, np.frombuffer(km, np.uint8), np.array([0, 31], np.uint64), np.array([0, 1], np.uint64))[0]
    regs = img[O.image_bytes(O.HLL, 16) - (1 << 16):]
    assert int(regs.max()) == 39 and int((regs != 0).sum()) == 1 and int(regs.argmax()) == 0x5fbf


def test_appendix_b_worked_example():
    seq = b"ACGTTGCATGCATCGATCGGATTACA"
    k16 = O.record_kmers(seq, 16)
    assert [hex(x) for x in k16[:3]] == ["0x1be4e4d8", "0x36393906", "0x8d8e4e41"]
    k21 = O.record_kmers(seq, 21)
    assert [hex(x) for x in k21[:3]] == ["0x6f93936368", "0xd636393906", "0x358d8e4e41"]
    assert len(k16) == 26 - 16 + 1 and len(k21) == 26 - 21 + 1
    # hashes (seed 42)
    lo, hi = O.xxh3_128_4b(0x1BE4E4D8, 42)
    assert (hi, lo) == (0xE278225F351F7BEA, 0xDE2D9BEC1FB5B282)
    assert O.xxh3_64_8b(0x1BE4E4D8, 42) == 0xF81686BDC4D6A8FD
    assert O.xxh3_64_8b(0x6F93936368, 42) == 0xEB51E070B7053DC7
    assert O.xxh3_64_8b(12345, 42) == 0x33D50D1BD6CF7FBC
    # register tuples for the first three k=16 k-mers, x = high64 hypothesis (switch U1 = 0)
    one = np.frombuffer(seq[:16], np.uint8)
    img = O.sketch_genomes(O.HMH, 16, 0, 42, one, np.array([0, 16], np.uint64), np.array([0, 1], np.uint64))[0]
    regs = img.view("<u2")
    assert regs[14494] == 0x1682 and np.count_nonzero(regs) == 1
    img = O.sketch_genomes(O.HMH, 16, 0, 42, one, np.array([0, 16], np.uint64), np.array([0, 1], np.uint64),
                           hmh_x_is_low=1)[0]
    regs = img.view("<u2")
    assert regs[14219] == 0x0BEA and np.count_nonzero(regs) == 1
    # HLL p=14: (j, rho) = (10493, 1); ULL p=12: idx 3969 first-hit register 4*(1+12-1) = 48
    img = O.sketch_genomes(O.HLL, 16, 14, 42, one, np.array([0, 16], np.uint64), np.array([0, 1], np.uint64))[0]
    assert img[33 + 10493] == 1 and np.count_nonzero(img[33:]) == 1
    assert img[24] == 14 and int(img[8:16].view("<u8")[0]) == 16383
    img = O.sketch_genomes(O.ULL, 16, 12, 42, one, np.array([0, 16], np.uint64), np.array([0, 1], np.uint64))[0]
    assert img[8 + 3969] == 48 and np.count_nonzero(img[8:]) == 1


def test_filter_and_mask():
    assert O.filter_out_n(b"ACGTNacgtRYKMACGT-*\n") == b"ACGTACGT"
    assert O.filter_out_n(b"") == b""
    assert O.mask_bits(2**64 - 1, 32) == 2**64 - 1
    assert O.mask_bits(2**64 - 1, 16) == 2**32 - 1
    assert O.mask_bits(2**64 - 1, 1) == 3
    assert O.mask_bits(0x123456789, 14) == 0x123456789 & (2**28 - 1)


@pytest.mark.parametrize("k", list(range(1, 33)))
def test_kmers_match_pyref_all_k(k):
    rng = random.Random(1000 + k)
    s = "".join(rng.choice("ACGTACGTACGTNacgtR") for _ in range(300))
    got = O.record_kmers(s.encode(), k)
    want = R.canonical_kmers(s, k)
    assert list(map(int, got)) == want
    # shorter than k after filtering -> nothing (utils.rs:460-462)
    short = ("ACGT" * 8)[:k - 1] + "NNNN"
    assert len(O.record_kmers(short.encode(), k)) == 0


def _as_arrays(records):
    seq = np.frombuffer(b"".join(records), np.uint8)
    off = np.cumsum([0] + [len(r) for r in records]).astype(np.uint64)
    return seq, off


@pytest.mark.parametrize("algo,k,p", [(O.HMH, 16, 0), (O.HMH, 21, 0), (O.HMH, 9, 0), (O.HLL, 21, 14), (O.HLL, 16, 4),
                                      (O.HLL, 32, 16), (O.ULL, 16, 12), (O.ULL, 15, 3), (O.ULL, 32, 9), (O.ULL, 17, 16)])
def test_sketch_matches_pyref(algo, k, p):
    rng = random.Random(7 * k + p)
    recs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(0, 700))) for _ in range(6)]
    recs[2] = recs[2][:40] + "NNNNN" + recs[2][40:].lower() + recs[2][40:]
    recs.append("ACGTAC")          # shorter than k for all k >= 7
    seq, off = _as_arrays([r.encode() for r in recs])
    img = O.sketch_genomes(algo, k, p, 42, seq, off, np.array([0, len(recs)], np.uint64))[0].tobytes()
    if algo == O.HMH:
        want = R.hmh_sketch(recs, k, 42)
    elif algo == O.HLL:
        want = R.hll_sketch(recs, k, p, 42)
    else:
        want = R.ull_sketch(recs, k, p, 42)
    assert img == want


def test_kmers_do_not_span_records_but_span_deleted_chars():
    a, b = b"ACGTACGTACGTACGTAAAA", b"CCCCGGGGTTTTACGTACGTAC"
    k = 16
    joined = O.record_kmers(a + b, k)
    sep = np.concatenate([O.record_kmers(a, k), O.record_kmers(b, k)])
    assert len(joined) == len(sep) + k - 1
    withn = O.record_kmers(a + b"NNnn" + b, k)       # deleted characters join the flanks (utils.rs:33-41)
    assert np.array_equal(withn, joined)


@pytest.mark.parametrize("algo,p", [(O.HMH, 0), (O.HLL, 12), (O.ULL, 10)])
def test_union_equals_sketch_of_concatenation(algo, p):
    rng = np.random.default_rng(5)
    recs = [bytes(rng.choice(list(b"ACGT"), size=3000).astype(np.uint8)) for _ in range(4)]
    seq, off = _as_arrays(recs)
    whole = O.sketch_genomes(algo, 16, p, 42, seq, off, np.array([0, 4], np.uint64))[0]
    parts = O.sketch_genomes(algo, 16, p, 42, seq, off, np.array([0, 1, 4], np.uint64))
    assert np.array_equal(O.merge_images(algo, p, parts[0], parts[1]), whole)
    assert np.array_equal(O.merge_images(algo, p, whole, whole), whole)      # idempotent


def test_parameter_errors():
    seq, off = _as_arrays([b"ACGT" * 20])
    g = np.array([0, 1], np.uint64)
    for algo, k, p in [(O.HMH, 0, 0), (O.HMH, 33, 0), (O.HLL, 16, 3), (O.HLL, 16, 17), (O.ULL, 16, 2), (O.ULL, 16, 27), (7, 16, 10)]:
        with pytest.raises(ValueError):
            O.sketch_genomes(algo, k, p, 42, seq, off, g)


def test_threads_and_order():
    gs = [O.synth_genome(g, 20000) for g in range(5)]
    seq, off = _as_arrays([g.tobytes() for g in gs])
    goff = np.arange(6, dtype=np.uint64)
    a = O.sketch_genomes(O.HMH, 16, 0, 42, seq, off, goff, threads=1)
    b = O.sketch_genomes(O.HMH, 16, 0, 42, seq, off, goff, threads=4)
    assert np.array_equal(a, b)
    assert len({hashlib.sha256(x.tobytes()).hexdigest() for x in a}) == 5


def test_synth_generator_golden():
    g0 = O.synth_genome(0, 100).tobytes()
    assert set(g0) <= set(b"ACGT")
    assert O.synth_genome(0, 1000)[:100].tobytes() == g0           # prefix-stable
    assert O.synth_genome(1, 100).tobytes() != g0
    assert hashlib.sha256(O.synth_genome(3, 50000).tobytes()).hexdigest()[:16] == SYNTH_G3_50K


SYNTH_G3_50K = "07a43913bec33b72"


def test_fixture_digests():
    want = json.load(open(os.path.join(GOLD, "fixture_digests.json")))
    algos = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}
    for key, dig in want.items():
        name, an, k, p, _ = key.split("|")
        recs = read_fastx(os.path.join(GOLD, name))
        seq, off = _as_arrays(recs)
        img = O.sketch_genomes(algos[an], int(k[1:]), int(p[1:]), 42, seq, off, np.array([0, len(recs)], np.uint64))[0]
        assert hashlib.sha256(img.tobytes()).hexdigest() == dig, key


# ---- the layout switches: every alternative the C oracle implements is cross-checked against the
# naive Python restatement, so that tools/ref_probe/fit_layout.py searches a space whose every point is independently stated
ALT_LAYOUTS = [
    dict(codes="ACTG"), dict(codes="TGCA", kmer="lsb"), dict(kmer="lsb"), dict(hmh_x="low", hmh_reg="be", hmh_hdr="l"),
    dict(hll_bucket="high", hll_hdr="pzsal"), dict(hll_hdr="PZ", ull_hdr="pL"), dict(ull_hdr="", hmh_hdr="Q", codes="GATC"),
]


@pytest.mark.parametrize("li", range(len(ALT_LAYOUTS)))
@pytest.mark.parametrize("algo,k,p", [(O.HMH, 16, 0), (O.HMH, 11, 0), (O.HMH, 27, 0), (O.HLL, 21, 10), (O.ULL, 16, 9), (O.ULL, 32, 5)])
def test_layout_alternatives_match_pyref(li, algo, k, p):
    kw = ALT_LAYOUTS[li]
    lay, rlay = O.make_layout(**kw), R.Lay(**kw)
    rng = random.Random(100 * li + k)
    recs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(0, 300))) for _ in range(4)]
    recs[1] = recs[1][:20] + "NnRY" + recs[1][20:]
    seq, off = _as_arrays([r.encode() for r in recs])
    img = O.sketch_genomes(algo, k, p, 42, seq, off, np.array([0, len(recs)], np.uint64), layout=lay)[0].tobytes()
    want = (R.hmh_sketch(recs, k, 42, lay=rlay) if algo == O.HMH else
            R.hll_sketch(recs, k, p, 42, lay=rlay) if algo == O.HLL else R.ull_sketch(recs, k, p, 42, lay=rlay))
    assert img == want
    assert len(img) == O.image_bytes(algo, p, lay)
    # union through the same layout
    parts = O.sketch_genomes(algo, k, p, 42, seq, off, np.array([0, 2, len(recs)], np.uint64), layout=lay)
    assert O.merge_images(algo, p, parts[0], parts[1], layout=lay).tobytes() == img
    assert O.parse_layout(lay.spec()).spec() == lay.spec()


def test_default_layout_is_the_survey_hypothesis():
    d = O.default_layout()
    assert d.spec() == "codes=ACGT,kmer=msb,hmh_x=high,hmh_reg=le,hll_bucket=low,hmh_hdr=,hll_hdr=azspl,ull_hdr=l,fastq_err=stop"
    assert O.header_bytes(O.HMH) == 0 and O.header_bytes(O.HLL) == 33 and O.header_bytes(O.ULL) == 8
    bad = O.make_layout()
    bad.base_code[0] = 1                                    # A and C share a code: not a permutation
    assert O.lib.lash_or_layout_check(bad) != 0
    with pytest.raises(ValueError):
        O.sketch_genomes(O.HMH, 16, 0, 42, np.frombuffer(b"ACGT" * 9, np.uint8), np.array([0, 36], np.uint64),
                         np.array([0, 1], np.uint64), layout=bad)


def test_file_buffers_equal_parsed_records():
    """lash_or_sketch_file_buffers (the oracle's needletail-like parse) == host parse (tests/fastx.py) + record path."""
    files = []
    for name in ("fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq"):
        files.append(open(os.path.join(GOLD, name), "rb").read())
    for algo, k, p in [(O.HMH, 16, 0), (O.HLL, 21, 12), (O.ULL, 31, 10)]:
        got = O.sketch_files(algo, k, p, 42, files, threads=3)
        for i, name in enumerate(("fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq")):
            recs = read_fastx(os.path.join(GOLD, name))
            seq, off = _as_arrays(recs)
            want = O.sketch_genomes(algo, k, p, 42, seq, off, np.array([0, len(recs)], np.uint64))[0]
            assert np.array_equal(got[i], want), (name, algo)
    with pytest.raises(ValueError):
        O.sketch_files(O.HMH, 16, 0, 42, [b"\n>late header\nACGT\n"])         # first byte must be '>' or '@'
    # a malformed FASTQ record ends the iteration; what came before is kept (utils.rs:457 `while let Some(Ok(..))`)
    good = b"@a\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n"
    broken = good + b"@b\nACGTACGTACGTACGTACGTAA\n+\nIII\n" + good.replace(b"ACGTACGT", b"TTTTGGGG")
    assert np.array_equal(O.sketch_files(O.HMH, 16, 0, 42, [broken]), O.sketch_files(O.HMH, 16, 0, 42, [good]))
