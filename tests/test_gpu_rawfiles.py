"""GPU tests of the device-side FASTA/FASTQ scan: lash_sketch_files_raw takes uncompressed file
bytes and must produce the same images as parsing the file on the host (needletail semantics, tests/fastx.py) and
sketching the records with the oracle."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O
from fastx import read_fastx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGO = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}


@pytest.fixture(scope="module")
def ctx():
    import lash_amd
    c = lash_amd.Context(0)
    yield c
    c.close()


def oracle_for_files(tmp_path, files_bytes, algo, k, p):
    out = []
    for i, fb in enumerate(files_bytes):
        path = tmp_path / ("f%d" % i)
        path.write_bytes(fb)
        recs = read_fastx(str(path))
        seq = np.frombuffer(b"".join(recs), np.uint8)
        off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
        out.append(O.sketch_genomes(ALGO[algo], k, p, 42, seq, off, np.array([0, len(recs)], np.uint64))[0])
    return np.stack(out)


def wrap(s, w, eol=b"\n"):
    return eol.join(s[i:i + w] for i in range(0, len(s), w)) + eol


def make_files():
    rng = random.Random(11)

    def rseq(n, alphabet="ACGT"):
        return "".join(rng.choice(alphabet) for _ in range(n)).encode()

    files = []
    for name in ("fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq"):
        files.append(open(os.path.join(GOLD, name), "rb").read())
    # multi-tile FASTA, 60-column lines, ACGT in the headers, N runs, lower case
    body = bytearray(rseq(700_000, "ACGT" * 20 + "Nacgt"))
    files.append(b">chr1 ACGTACGTACGT description with bases GATTACA\n" + wrap(bytes(body), 60) +
                 b">chr2 TTTT\n" + wrap(rseq(123_457), 70) + b">empty\n>tiny ACGT\nACGTAC\n")
    # single-line records: no newline inside 300 kb of sequence; a header longer than one 16 KiB tile
    files.append(b">" + b"ACGT" * 6000 + b" long header\n" + rseq(300_000) + b"\n>second\n" + rseq(50_000))
    # CRLF line ends, no trailing newline
    files.append(b">crlf\r\n" + wrap(rseq(40_000), 80, b"\r\n") + b">last\r\n" + rseq(33))
    # FASTQ: 150-bp reads with ACGT-looking quality strings and '@' / '>' inside qualities
    reads = []
    for i in range(3000):
        r = rseq(rng.choice([150, 150, 150, 76, 12]), "ACGT" * 30 + "N")
        qual = bytes(rng.choice(b"ACGT@>+IIIIFFFF#") for _ in range(len(r)))
        reads.append(b"@read%d ACGT\n" % i + r + b"\n+\n" + qual + b"\n")
    files.append(b"".join(reads))
    files.append(b"@only\nACGTACGTACGTACGTACGTA\n+\nIIIIIIIIIIIIIIIIIIIII")        # no trailing newline
    files.append(b">no sequence at all\n")
    return files


@pytest.mark.parametrize("algo,k,p", [("hmh", 16, 0), ("hll", 21, 12), ("ull", 16, 12), ("hmh", 7, 0)])
def test_raw_files_match_host_parse_plus_oracle(ctx, tmp_path, algo, k, p):
    files = make_files()
    got = ctx.sketch_files_raw(algo, k, p, 42, files)
    want = oracle_for_files(tmp_path, files, algo, k, p)
    for i in range(len(files)):
        assert np.array_equal(got[i], want[i]), "file %d (%d bytes) differs for %s" % (i, len(files[i]), algo)


def test_raw_large_wrapped_fasta(ctx, tmp_path):
    """A 20 Mbp genome as 80-column FASTA: ~1 250 tiles, every tile starts mid-line; plus many files in one call."""
    g = O.synth_genome(4242, 20_000_000).tobytes()
    files = [b">big\n" + wrap(g, 80)] + [b">g%d\n" % i + wrap(O.synth_genome(5000 + i, 300_000 + 17 * i).tobytes(), 61) for i in range(40)]
    got = ctx.sketch_files_raw("hmh", 16, 0, 42, files)
    want0 = O.sketch_genomes(O.HMH, 16, 0, 42, np.frombuffer(g, np.uint8), np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0]
    assert np.array_equal(got[0], want0)
    want = oracle_for_files(tmp_path, files[1:], "hmh", 16, 0)
    assert np.array_equal(got[1:], want)


def test_header_only_where_a_line_opens(ctx, tmp_path):
    """needletail finds the next FASTA record at "\\n>": a '>' in the middle of a line is sequence text (deleted by the
    ACGT filter, so its flanks join).  Checked at every position of a 16-byte lane chunk and across a tile boundary."""
    g = O.synth_genome(66, 40_000).tobytes()
    files = []
    for shift in range(0, 18):
        body = g[:100 + shift] + b">not a header ACGTACGT" + g[100 + shift:300] + b"\n" + g[300:16_390 - shift] + b">" + g[16_390 - shift:20_000]
        files.append(b">r0\n" + body + b"\n>r1\n" + g[20_000:20_100] + b"\n")
    files.append(b">a\nACGT>b\nACGT\n")                     # one record: the flanks of the deleted text join
    (tmp_path / "one").write_bytes(files[-1])
    assert read_fastx(str(tmp_path / "one")) == [b"ACGT>bACGT"]
    got = ctx.sketch_files_raw("hmh", 16, 0, 42, files)
    assert np.array_equal(got, oracle_for_files(tmp_path, files, "hmh", 16, 0))
    got = ctx.sketch_files_raw("ull", 4, 8, 42, files)
    assert np.array_equal(got, oracle_for_files(tmp_path, files, "ull", 4, 8))


def _fastq(reads, qual=b"I"):
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, qual * len(r)) for i, r in enumerate(reads))


def test_malformed_fastq_is_detected_on_the_device_and_redone_exactly(ctx):
    """the device FASTQ parse is "newlines so far mod 4"; a broken record would mis-phase everything after it and
    quality lines (ACGT are valid Phred characters) would be hashed as sequence.  Now: a line in phase 0 that does not start
    with '@', or one in phase 2 that does not start with '+', flags the file; the host-buffer entry re-does a flagged file
    with needletail's rule — iteration stops at the malformed record, what came before stands (utils.rs:457) — and the
    device-buffer entry reports LASH_EFORMAT.  Well-formed files in the same call are untouched."""
    import lash_amd
    import torch
    rng = random.Random(5)
    reads = ["".join(rng.choice("ACGT") for _ in range(rng.randint(30, 300))).encode() for _ in range(400)]
    good = _fastq(reads, b"A")                                   # qualities that look like sequence
    cases = {
        "blank line inside": _fastq(reads[:100], b"C") + b"\n" + _fastq(reads[100:], b"G"),
        "missing plus": _fastq(reads[:57], b"T") + b"@x\nACGTACGTACGTACGTACGTACGT\nACGTACGTACGTACGTACGTACGT\n" + _fastq(reads[57:], b"A"),
        "header without @": _fastq(reads[:200]) + b"r200\n" + reads[200] + b"\n+\n" + b"I" * len(reads[200]) + b"\n" + _fastq(reads[201:]),
        "first record broken": b"@a\nACGTACGTACGTACGTACGTTTGA\nIIIIIIIIIIIIIIIIIIIIIIII\n" + good,
        "trailing blank lines": good + b"\n\n",
        "quality starts with @ and +": _fastq(reads[:50], b"@") + _fastq(reads[50:120], b"+") + _fastq(reads[120:]),   # well-formed!
        "crlf": good.replace(b"\n", b"\r\n"),                       # well-formed
    }
    names = list(cases)
    files = [cases[n] for n in names] + [good]
    for an, k, p in (("hmh", 16, 0), ("ull", 21, 10)):
        want = O.sketch_files(ALGO[an], k, p, 42, files, threads=4)     # the oracle's parse stops at the first malformed record
        got = ctx.sketch_files_raw(an, k, p, 42, files)
        bad = ctx.format_errors()
        for i, n in enumerate(names + ["good"]):
            assert np.array_equal(got[i], want[i]), (an, n)
        flagged = {names[i] for i in bad if i < len(names)}
        assert flagged == {"blank line inside", "missing plus", "header without @", "first record broken", "trailing blank lines"}, flagged
        # accumulate into existing images takes the same path
        acc = ctx.sketch_files_raw(an, k, p, 42, [good] * len(files))
        prm_files = files
        raw = np.frombuffer(b"".join(prm_files), np.uint8)
        off = np.cumsum([0] + [len(f) for f in prm_files]).astype(np.uint64)
        fmt = np.full(len(prm_files), 2, np.uint8)
        import ctypes as C
        from lash_amd import _lib
        prm = _lib.Params(lash_amd.ALGOS[an], k, p, lash_amd.F_ACCUMULATE, 42)
        rc = _lib.load().lash_sketch_files_raw(ctx._h, C.byref(prm), raw.ctypes.data, off.ctypes.data, fmt.ctypes.data, len(prm_files), acc.ctypes.data)
        assert rc == 0
        for i in range(len(files)):
            assert np.array_equal(acc[i], O.merge_images(ALGO[an], p, want[-1], want[i])), (an, i)
    # device-buffer entry: no host copy to fall back to -> LASH_EFORMAT at synchronize, indices reported
    d_raw = torch.from_numpy(np.frombuffer(b"".join(files), np.uint8).copy()).cuda()
    off = np.cumsum([0] + [len(f) for f in files]).astype(np.uint64)
    fmt = np.full(len(files), 2, np.uint8)
    d_img = torch.zeros(len(files) * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device="cuda")
    import ctypes as C
    from lash_amd import _lib
    prm = _lib.Params(0, 16, 0, 0, 42)
    rc = _lib.load().lash_sketch_files_raw_device(ctx._h, C.byref(prm), d_raw.data_ptr(), off.ctypes.data, fmt.ctypes.data, len(files), d_img.data_ptr())
    assert rc == 0
    with pytest.raises(lash_amd.LashError) as e:
        ctx.synchronize()
    assert e.value.code == _lib.EFORMAT
    assert {names[i] for i in ctx.format_errors()} == {"blank line inside", "missing plus", "header without @", "first record broken", "trailing blank lines"}
    img = d_img.cpu().numpy().reshape(len(files), -1)
    want = O.sketch_files(O.HMH, 16, 0, 42, files, threads=4)
    for i, n in enumerate(names + ["good"]):
        if i not in ctx.format_errors():
            assert np.array_equal(img[i], want[i]), n              # the well-formed ones are right
    ctx.synchronize()                                              # the error was consumed
    # a quality line whose length differs from its sequence line's is needletail's other FASTQ error (the reference stops there).
    # The 4-line structure is intact, so the line-start check above cannot see it; fastq_check.hip compares the line lengths on
    # the device (round 3) and the file is re-done exactly like the others.
    short_q = _fastq(reads[:10]) + b"@x\n" + reads[10] + b"\n+\n" + b"I" * (len(reads[10]) - 1) + b"\n" + _fastq(reads[11:])
    got = ctx.sketch_files_raw("hmh", 16, 0, 42, [good, short_q])
    assert ctx.format_errors() == [1]
    assert np.array_equal(got, O.sketch_files(O.HMH, 16, 0, 42, [good, short_q]))     # the oracle (needletail's rule) stops at record 10
    assert not np.array_equal(got[1], O.sketch_files(O.HMH, 16, 0, 42, [_fastq(reads)])[0])
    with pytest.raises(lash_amd.LashError):                        # first byte rule (parse_fastx_file fails, utils.rs:453)
        ctx.sketch_files_raw("hmh", 16, 0, 42, [b"\n" + good])


def _device_flags(ctx, files):
    """lash_sketch_files_raw_device on FASTQ files -> the set of flagged file indices"""
    import ctypes as C
    import lash_amd
    import torch
    from lash_amd import _lib
    d_raw = torch.from_numpy(np.frombuffer(b"".join(files) + b"\0" * 64, np.uint8).copy()).cuda()
    off = np.cumsum([0] + [len(f) for f in files]).astype(np.uint64)
    fmt = np.full(len(files), 2, np.uint8)
    d_img = torch.zeros(len(files) * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device="cuda")
    prm = _lib.Params(0, 16, 0, 0, 42)
    rc = _lib.load().lash_sketch_files_raw_device(ctx._h, C.byref(prm), d_raw.data_ptr(), off.ctypes.data, fmt.ctypes.data, len(files), d_img.data_ptr())
    assert rc == 0
    try:
        ctx.synchronize()
    except lash_amd.LashError as e:
        assert e.code == _lib.EFORMAT
    return set(ctx.format_errors()), d_img.cpu().numpy().reshape(len(files), -1)


def _host_says_bad(f):
    from lash_amd import _lib
    buf = np.frombuffer(f, np.uint8)
    return int(_lib.load().lash_fastq_valid_prefix(buf.ctypes.data, len(f))) < len(f)


def test_fastq_quality_length_check_on_the_device_is_exact(ctx):
    """the device-buffer entry flags EXACTLY the files needletail's rule (the host parse, lash_fastq_valid_prefix)
    stops in — quality lines of a different length anywhere in the file, lines that span several 4 KiB blocks of the check,
    CRLF, a last record without its final newline, a file cut inside a record — and no well-formed file."""
    rng = random.Random(77)

    def read(n):
        return "".join(rng.choice("ACGT") for _ in range(n)).encode()

    def rec(i, s, q=None, nl=b"\n"):
        return b"@r%d" % i + nl + s + nl + b"+" + nl + (b"I" * len(s) if q is None else q) + nl

    short = [read(rng.randint(20, 300)) for _ in range(300)]
    longr = [read(rng.choice([4000, 4096, 4097, 9000, 20000, 70000])) for _ in range(12)]
    ok_short = b"".join(rec(i, s) for i, s in enumerate(short))
    ok_long = b"".join(rec(i, s) for i, s in enumerate(longr))
    files = {
        "ok short": ok_short,
        "ok long": ok_long,
        "ok crlf": b"".join(rec(i, s, nl=b"\r\n") for i, s in enumerate(short)),
        "ok no final newline": ok_short[:-1],
        "ok long no final newline": ok_long[:-1],
        "ok empty read": rec(0, b"") + rec(1, short[1]) + rec(2, b""),
        "ok one record": rec(0, short[0]),
        "ok empty last quality, no newline": rec(0, short[0]) + b"@e\n\n+\n",       # the host parse takes it (no base in it anyway)
        "ok cr only in quality": rec(0, b"ACGT", b"IIII\r") + rec(1, b"ACGT\r", b"IIII"),
        "bad short quality": b"".join(rec(i, s, b"I" * (len(s) - 1) if i == 150 else None) for i, s in enumerate(short)),
        "bad long quality (last record)": b"".join(rec(i, s, b"I" * (len(s) + 1) if i == 299 else None) for i, s in enumerate(short)),
        "bad first record": b"".join(rec(i, s, b"" if i == 0 else None) for i, s in enumerate(short)),
        "bad long read, quality one short": b"".join(rec(i, s, b"I" * (len(s) - 1) if i == 7 else None) for i, s in enumerate(longr)),
        "bad long read, quality a block longer": b"".join(rec(i, s, b"I" * (len(s) + 4096) if i == 3 else None) for i, s in enumerate(longr)),
        "bad last quality short, no newline": ok_short[:-2],
        "bad cut after plus line": ok_short + b"@x\nACGT\n+\n",
        "bad cut after sequence": ok_short + b"@x\nACGT\n",
        "bad cut in header": ok_short + b"@x",
        "bad crlf quality short": b"".join(rec(i, s, b"I" * (len(s) - 1) if i == 20 else None, nl=b"\r\n") for i, s in enumerate(short)),
        "bad swapped lengths": rec(0, b"ACGTACGT", b"IIII") + rec(1, b"ACGT", b"IIIIIIII"),
    }
    names = list(files)
    data = [files[n] for n in names]
    want_bad = {i for i, f in enumerate(data) if _host_says_bad(f)}
    assert {names[i] for i in want_bad} == {n for n in names if n.startswith("bad")}        # the host rule itself, as a sanity check
    flagged, img = _device_flags(ctx, data)
    assert {names[i] for i in flagged} == {names[i] for i in want_bad}
    want = O.sketch_files(O.HMH, 16, 0, 42, data, threads=4)
    for i in range(len(data)):
        if i not in flagged:
            assert np.array_equal(img[i], want[i]), names[i]
    # and the host-buffer entry re-does the flagged ones exactly
    got = ctx.sketch_files_raw("hmh", 16, 0, 42, data)
    assert np.array_equal(got, want)
    # randomized: files of random records, one random defect (or none) each
    for it in range(30):
        data = []
        for _ in range(rng.randint(1, 8)):
            nl = rng.choice([b"\n", b"\r\n"])
            recs = []
            for i in range(rng.randint(1, 120)):
                s = read(rng.choice([0, 1, 50, 150, rng.randint(1, 600), rng.choice([4090, 5000, 13000])]))
                recs.append([b"@r%d" % i, s, b"+", bytes(rng.choice(b"I@+ACGT#") for _ in range(len(s)))])
            if rng.random() < 0.5:
                r = rng.choice(recs)
                d = rng.choice([-1, 1, 2, -len(r[1]) // 2, 4096])
                r[3] = r[3][:len(r[3]) + d] if d < 0 else r[3] + b"I" * d
            f = b"".join(nl.join(r) + nl for r in recs)
            if rng.random() < 0.3:
                f = f[:-len(nl)]
            if rng.random() < 0.15:
                f = f[:rng.randint(1, len(f))]
            data.append(f)
        want_bad = {i for i, f in enumerate(data) if _host_says_bad(f)}
        flagged, img = _device_flags(ctx, data)
        assert flagged == want_bad, (it, sorted(flagged), sorted(want_bad))
