"""CPU tests of the C++ host side (lash_amd/csrc/host): FASTX reader, list-file rules, JSON writers, zstd stream,
and the `lash` command line's behaviour without a GPU."""
import bz2
import gzip
import lzma
import json
import os
import subprocess

import numpy as np
import pytest

import host_lib as H
from fastx import read_fastx as py_read_fastx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq"]


@pytest.mark.parametrize("name", FIXTURES)
def test_fastx_reader_matches_needletail_semantics(name, tmp_path):
    path = os.path.join(GOLD, name)
    want = py_read_fastx(path)
    assert H.read_fastx(path) == want
    gz = tmp_path / (name + ".gz")                       # needletail sniffs the gzip magic
    with open(path, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    assert H.read_fastx(str(gz)) == want
    zs = tmp_path / (name + ".zst")
    H.zstd_write(str(zs), open(path, "rb").read())
    assert H.read_fastx(str(zs)) == want
    raw = open(path, "rb").read()
    bz = tmp_path / (name + ".bz2")                      # needletail's default features also inflate bzip2 and xz
    bz.write_bytes(bz2.compress(raw))
    assert H.read_fastx(str(bz)) == want
    xz = tmp_path / (name + ".xz")
    xz.write_bytes(lzma.compress(raw, format=lzma.FORMAT_XZ))
    assert H.read_fastx(str(xz)) == want


def test_bzip2_and_xz_streams_large_concatenated_and_corrupt(tmp_path):
    rec = b">r%d\n" + b"ACGTTGCA" * 4000 + b"\n"
    parts = [rec % i for i in range(40)]                  # 1.3 MB: several refills of the 1 MiB input buffer
    want = [b"ACGTTGCA" * 4000] * 40
    p = tmp_path / "m.fa.bz2"
    p.write_bytes(b"".join(bz2.compress(b"".join(parts[i:i + 10])) for i in range(0, 40, 10)))   # 4 concatenated streams
    assert H.read_fastx(str(p)) == want
    p = tmp_path / "m.fa.xz"
    p.write_bytes(b"".join(lzma.compress(b"".join(parts[i:i + 20]), format=lzma.FORMAT_XZ) for i in range(0, 40, 20)))
    assert H.read_fastx(str(p)) == want
    good = bz2.compress(b"".join(parts))
    p = tmp_path / "t.fa.bz2"
    p.write_bytes(good[:len(good) // 2])
    with pytest.raises(ValueError, match="bzip2"):
        H.read_fastx(str(p))
    good = lzma.compress(b"".join(parts), format=lzma.FORMAT_XZ)
    p = tmp_path / "t.fa.xz"
    p.write_bytes(good[:len(good) // 2] + b"\x00" * 64)
    with pytest.raises(ValueError, match="xz"):
        H.read_fastx(str(p))


def test_fastx_reader_edge_cases(tmp_path):
    p = tmp_path / "e.fa"
    p.write_bytes(b">only header\n")
    assert H.read_fastx(str(p)) == [b""]
    p.write_bytes(b">a\nACGT\n\nTT\n>b\n>c\nGG")
    assert H.read_fastx(str(p)) == [b"ACGTTT", b"", b"GG"]
    p.write_bytes(b"")
    assert H.read_fastx(str(p)) == []
    p.write_bytes(b"@r1\nACGTN\n+\nIIIII\n@r2\nGGCC\n+r2\nIIII\n")
    assert H.read_fastx(str(p)) == [b"ACGTN", b"GGCC"]
    p.write_bytes(b"not a sequence file\n")
    with pytest.raises(ValueError, match="Invalid input file"):
        H.read_fastx(str(p))
    with pytest.raises(ValueError, match="Invalid input file"):
        H.read_fastx(str(tmp_path / "missing.fa"))


def test_list_file_rules(tmp_path):
    # main.rs:200-207: lines().filter(|l| !l.trim().is_empty()) — kept lines are NOT trimmed
    p = tmp_path / "list.txt"
    p.write_bytes(b"a.fa\n\n  \n b.fa \r\nc.fa")
    assert H.read_list(str(p)) == ["a.fa", " b.fa ", "c.fa"]


def test_json_documents_match_serde_pretty(tmp_path):
    assert H.json_array([]) == "[]"
    assert H.json_array(["a.fa", 'we"ird\\name\t.fa']) == '[\n  "a.fa",\n  "we\\"ird\\\\name\\t.fa"\n]'
    assert json.loads(H.json_array(["x", "y/z.fasta"])) == ["x", "y/z.fasta"]
    pre = str(tmp_path / "out")
    H.write_parameters(pre, "hmh", 16, 10, 42)
    assert open(pre + "_parameters.json").read() == \
        '{\n  "algorithm": "hmh",\n  "k": "16",\n  "molecule": "nucleotide",\n  "seed": "42"\n}'
    H.write_parameters(pre, "ull", 21, 12, 7)
    assert open(pre + "_parameters.json").read() == \
        '{\n  "algorithm": "ull",\n  "k": "21",\n  "molecule": "nucleotide",\n  "precision": "12",\n  "seed": "7"\n}'


def test_zstd_stream_roundtrip_and_interop(tmp_path):
    rng = np.random.default_rng(3)
    data = rng.integers(0, 4, size=3_000_000, dtype=np.uint8).tobytes() + b"\x00" * 100_000
    p = str(tmp_path / "x.bin")
    H.zstd_write(p, data, level=3, workers=4)
    assert os.path.getsize(p) < len(data)
    assert H.zstd_read(p) == data
    try:
        import pyarrow as pa
    except Exception:
        return
    raw = open(p, "rb").read()
    assert raw[:4] == b"\x28\xb5\x2f\xfd"               # one standard zstd frame: any decoder (incl. the zstd crate) reads it
    assert pa.decompress(raw, decompressed_size=len(data), codec="zstd").to_pybytes() == data


def test_zstd_frames_from_several_threads(tmp_path):
    """A libzstd without multithreading (this image's): the writer cuts the stream into 4 MiB pieces that its threads compress as
    independent frames (zstd_dl.hpp).  The file is a concatenation of frames — what any zstd decoder reads as one stream; the bytes that
    come out are the bytes that went in, for streams of no, one and a few bytes, of exactly one piece, of exactly one round of the threads, and beyond."""
    rng = np.random.default_rng(4)
    big = (rng.integers(0, 16, size=37_000_001, dtype=np.uint8)).tobytes()
    for data in (b"", b"x", big[:(4 << 20)], big[:(16 << 20)], big[:(16 << 20) + 1], big):
        p = str(tmp_path / "y.bin")
        H.zstd_write(p, data, level=3, workers=4)
        assert H.zstd_read(p) == data
        raw = open(p, "rb").read()
        assert raw[:4] == b"\x28\xb5\x2f\xfd"
        if len(data) > (16 << 20):
            assert raw.count(b"\x28\xb5\x2f\xfd") >= 4           # several frames
        try:
            import pyarrow as pa
        except Exception:
            continue
        if data:
            assert pa.decompress(raw, decompressed_size=len(data), codec="zstd").to_pybytes() == data


def test_cli_without_gpu_fails_loudly(tmp_path):
    import lash_amd
    if lash_amd.load().lash_device_count() > 0:
        pytest.skip("a GPU is present")
    lst = tmp_path / "l.txt"
    lst.write_text(os.path.join(GOLD, "fixture_A.fasta") + "\n")
    r = subprocess.run([H.CLI, "sketch", "-f", str(lst), "-o", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "no usable HIP device" in r.stderr
    assert "initializing logger" in r.stdout            # main.rs:23 banner
    r = subprocess.run([H.CLI, "sketch", "-f", str(lst), "-a", "minhash"], capture_output=True, text=True)
    assert r.returncode != 0 and "Algorithm must be either hmh, ull, or hll" in r.stderr
    r = subprocess.run([H.CLI, "sketch"], capture_output=True, text=True)
    assert r.returncode == 2 and "--file" in r.stderr


def _kmer_set(path_bytes, k, tmp_path, name):
    """canonical k-mers of a FASTA/FASTQ byte string, via the host reader (needletail semantics) + the oracle"""
    import oracle_lib as O
    p = tmp_path / name
    p.write_bytes(path_bytes)
    out = set()
    for rec in H.read_fastx(str(p)):
        out.update(int(x) for x in O.record_kmers(rec, k))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_streaming_chunks_cover_exactly_the_kmers_of_the_file(tmp_path, seed):
    """The large-file streamer cuts a file into chunks (`stream_find_cut`) and sketches them into one image.  Property: the
    union of the chunks' k-mer sets equals the file's, whatever the record sizes, line widths and N runs at the cuts
    (a record larger than a chunk is cut at a line end, or inside one enormous line, and the last <= 32 surviving bases are
    carried over)."""
    import random
    import oracle_lib as O
    rng = random.Random(seed)
    g = O.synth_genome(900 + seed, 400_000).tobytes()
    parts, pos = [], 0
    while pos < 300_000:
        n = rng.choice([rng.randint(1, 300), rng.randint(5_000, 40_000), rng.randint(60_000, 120_000)])
        s = bytearray(g[pos:pos + n])
        if rng.random() < 0.6 and len(s) > 10:
            i = rng.randrange(len(s))
            run = rng.choice([1, 40, 3_000, 30_000])
            s[i:i + run] = b"N" * len(s[i:i + run])
        w = rng.choice([60, 80, 10**9, 7])
        parts.append(b">r%d\n" % len(parts) + b"\n".join(bytes(s[i:i + w]) for i in range(0, len(s), w)) + b"\n")
        pos += n
    data = b"".join(parts)
    chunk = rng.choice([20_000, 33_333, 50_000])
    for k in (16, 31):
        want = _kmer_set(data, k, tmp_path, "whole.fa")
        got, buf, rest = set(), b"", data
        while rest or buf:
            take = chunk - len(buf)
            buf, rest = buf + rest[:take], rest[take:]
            if rest:
                cut, carry = H.stream_find_cut(buf, 1)
                assert len(buf) // 2 < cut <= len(buf) and len(carry) <= 33
            else:
                cut, carry = len(buf), b""
            piece = buf[:cut]
            got |= _kmer_set(piece if piece.startswith(b">") else b">continued\n" + piece, k, tmp_path, "piece.fa")
            buf = carry + buf[cut:]
        assert got == want, (seed, k, len(got ^ want))


@pytest.mark.parametrize("seed", range(4))
def test_streaming_chunks_of_fastq_split_between_records(tmp_path, seed):
    """FASTQ chunks end just before a record start: an '@' that opens a line whose line-after-next starts with '+'
    (quality lines may start with '@' or '+').  The records of the chunks, in order, are the records of the file."""
    import random
    rng = random.Random(100 + seed)
    recs = []
    for i in range(3000):
        L = rng.choice([1, 36, 100, 150, rng.randint(1, 300)])
        s = bytes(rng.choice(b"ACGTN") for _ in range(L))
        q = bytes(rng.choice(b"@+I#F>A") for _ in range(L))
        recs.append(b"@r%d %s\n" % (i, rng.choice([b"", b"@+x"])) + s + b"\n+" + rng.choice([b"", b"r%d" % i]) + b"\n" + q + b"\n")
    data = b"".join(recs)
    (tmp_path / "w.fq").write_bytes(data)
    want = H.read_fastx(str(tmp_path / "w.fq"))
    assert len(want) == 3000
    chunk = rng.choice([4_000, 10_007, 50_000])
    got, buf, rest = [], b"", data
    while rest or buf:
        take = chunk - len(buf)
        buf, rest = buf + rest[:take], rest[take:]
        if rest:
            cut, carry = H.stream_find_cut(buf, 2)
            assert carry == b"" and len(buf) // 2 < cut <= len(buf)
        else:
            cut = len(buf)
        (tmp_path / "p.fq").write_bytes(buf[:cut])
        got += H.read_fastx(str(tmp_path / "p.fq"))
        buf = buf[cut:]
    assert got == want


def test_json_array_matches_a_reference_pretty_printer_on_odd_names():
    """File names go into {o}_files.json verbatim (utils.rs:577-580, serde_json::to_string_pretty).  Python's json module
    with indent=2 / ensure_ascii=False writes the same text as serde_json for arrays of strings: same escapes
    (\\" \\\\ \\b \\f \\n \\r \\t, \\u00XX for the other control characters), non-ASCII left as UTF-8."""
    import random
    rng = random.Random(5)
    alphabet = [chr(c) for c in list(range(1, 10)) + list(range(11, 128))] + ["é", "λ", "雪", "𝄞", '"', "\\", "/", "\t", "\x7f"]
    for _ in range(200):
        items = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 30))) for _ in range(rng.randint(0, 6))]
        got = H.json_array(items)
        assert got == json.dumps(items, indent=2, ensure_ascii=False) or (items == [] and got == "[]")
        assert json.loads(got) == items


def test_parallel_multi_member_gzip_reader(tmp_path, monkeypatch):
    """pgzip.hpp: members of a concatenated .gz inflate in parallel, speculatively from candidate magic bytes; a result is used
    only when the previous member ended exactly there.  Whatever the file looks like, the bytes must equal gzip's."""
    import gzip
    import zlib
    rng = np.random.default_rng(8)

    def blob(n):
        return bytes(np.frombuffer(b"ACGTN\n@+I", np.uint8)[rng.integers(0, 9, size=n)])
    magic = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\x03"             # a plausible member header as PAYLOAD: false candidates
    parts = [blob(300_000), b"", blob(17), magic * 50 + blob(100_000) + magic, blob(1_500_000), magic, blob(64)]
    cases = {
        "many members": b"".join(gzip.compress(p, 1) for p in parts),
        "single member": gzip.compress(b"".join(parts), 6),
        # stored (level 0) members keep the fake headers verbatim in the compressed stream
        "false candidates inside stored members": b"".join(gzip.compress(p, 0) for p in parts),
        # a member header with FNAME + FCOMMENT (flags 0x18): the scan accepts any reserved-bit-free flag byte
        "named members": b"".join(_gz_named(p, b"lane%d.fastq" % i) for i, p in enumerate(parts)),
        "empty file": b"",
    }
    want_all = b"".join(parts)
    for name, data in cases.items():
        f = tmp_path / (name.replace(" ", "_") + ".gz")
        f.write_bytes(data)
        want = want_all if data else b""
        for threads in (1, 2, 8):
            for read_size in (1 << 20, 4097):
                got, n_par, n_seq = H.pgzip_read(str(f), threads, read_size)
                assert got == want, (name, threads, read_size)
                if threads > 1 and name == "many members":
                    assert n_par >= 5, (n_par, n_seq)
                if threads == 1:
                    assert n_par == 0
    # a member larger than the per-member buffer is handed to the sequential path (no unbounded buffering)
    monkeypatch.setenv("LASH_PGZIP_MEMBER_CAP", "200000")
    # (the cap is read once per process: exercise it in a fresh one)
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import host_lib as H; d, p, s = H.pgzip_read(%r, 4); "
            "import gzip; assert d == gzip.open(%r).read(); print(p, s)" % (os.path.dirname(os.path.abspath(__file__)),
                                                                         str(tmp_path / "many_members.gz"), str(tmp_path / "many_members.gz")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, LASH_PGZIP_MEMBER_CAP="200000"))
    assert r.returncode == 0, r.stderr
    n_par, n_seq = map(int, r.stdout.split())
    assert n_seq >= 2 and n_par >= 3                              # the 300 kB and 1.5 MB members went sequentially
    # corruption and truncation are errors, not silence
    good = cases["many members"]
    (tmp_path / "trunc.gz").write_bytes(good[:len(good) - 7])
    bad = bytearray(good)
    bad[len(gzip.compress(parts[0], 1)) + 5000] ^= 0x55
    (tmp_path / "corrupt.gz").write_bytes(bytes(bad))
    for name in ("trunc.gz", "corrupt.gz"):
        for threads in (1, 4):
            with pytest.raises(ValueError):
                H.pgzip_read(str(tmp_path / name), threads)


def _gz_named(payload, fname):
    import struct
    import zlib
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = co.compress(payload) + co.flush()
    hdr = b"\x1f\x8b\x08\x18" + b"\x00\x00\x00\x00" + b"\x04\x03" + fname + b"\x00" + b"a comment\x00"
    return hdr + body + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF)


def test_fastq_valid_prefix_is_needletails_rule():
    """lash_fastq_valid_prefix (C ABI, host only): where needletail's FASTQ iterator stops == where the oracle's parse stops."""
    import ctypes as C
    import lash_amd
    import oracle_lib as O
    lib = lash_amd.load()
    rec = lambda i, s, q=None: b"@r%d\n%s\n+\n%s\n" % (i, s, (b"I" * len(s)) if q is None else q)
    good = b"".join(rec(i, b"ACGT" * (5 + i)) for i in range(20))
    cases = [
        (good, len(good)),
        (good[:-1], len(good) - 1),                                              # no final newline: still a record
        (good + b"\n", len(good)),                                               # trailing blank line: stops there
        (good + rec(99, b"ACGTACGT", b"III"), len(good)),                        # quality shorter than sequence
        (good + b"@x\nACGT\n-\nIIII\n" + good, len(good)),                       # '+' line missing
        (good + b"x\nACGT\n+\nIIII\n", len(good)),                               # header without '@'
        (good + b"@trunc\nACGT\n+\n", len(good)),                                # file ends inside a record
        (good.replace(b"\n", b"\r\n"), len(good.replace(b"\n", b"\r\n"))),       # CRLF
        (b"", 0), (b">fasta\nACGT\n", 0), (b"\n" + good, 0),
    ]
    for data, want in cases:
        buf = np.frombuffer(data, np.uint8).copy() if data else np.zeros(1, np.uint8)
        got = int(lib.lash_fastq_valid_prefix(buf.ctypes.data, len(data)))
        assert got == want, (data[-40:], got, want)
        if data[:1] == b"@":
            # the oracle sketches exactly that prefix
            a = O.sketch_files(O.HMH, 16, 0, 42, [data])
            b = O.sketch_files(O.HMH, 16, 0, 42, [data[:got]]) if got else O.sketch_files(O.HMH, 16, 0, 42, [b"@e\n\n+\n\n"])
            assert np.array_equal(a, b)
            # and the neutralised buffer is well-formed to its end, with the same records
            if got < len(data):
                lib.lash_fastq_neutralise_tail(buf.ctypes.data + got, len(data) - got)
                fixed = buf.tobytes()[:len(data)]
                if len(data) - got >= 6:                         # room for a whole (empty-sequence) record
                    assert int(lib.lash_fastq_valid_prefix(buf.ctypes.data, len(data))) == len(data), fixed[got:got + 20]
                else:                                            # a record cut short by the end of the file: '@' and '+' lines still in phase
                    assert fixed[got:] == b"@\n\n+\n"[:len(data) - got]
                assert np.array_equal(O.sketch_files(O.HMH, 16, 0, 42, [fixed]), a)


def test_xxh3_64_seeded_equals_python_xxhash():
    """name_order.cpp's XXH3-64 against the python-xxhash module (an independent binding of the C library): every
    length class (0, 1-3, 4-8, 9-16, 17-128, 129-240, > 240 incl. block boundaries) and the seeds that matter."""
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(11)
    for n in list(range(0, 300)) + [511, 512, 513, 1023, 1024, 1025, 1087, 1088, 1089, 2048, 2049, 4097, 10_000]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 93, 42, 2**64 - 1, 0x0123456789ABCDEF):
            assert H.xxh3_64(data, seed) == xxhash.xxh3_64_intdigest(data, seed=seed), (n, seed)


def test_name_order_equals_the_control_byte_restatement():
    """The host's simplified table walk against tests/pyref.py's hashbrown restatement that keeps the control bytes,
    group loads and fix_insert_slot: every table size from the empty singleton to 4096 buckets, with repeated names."""
    pytest.importorskip("xxhash")
    import pyref as R
    rng = np.random.default_rng(12)
    for n in [0, 1, 2, 3, 4, 5, 7, 8, 9, 14, 15, 16, 17, 28, 29, 30, 56, 57, 58, 112, 113, 300, 1000, 3500]:
        for rep in range(3):
            names = ["dir%d/genome_%d.fna.gz" % (int(rng.integers(0, 50)), int(rng.integers(0, 10**9))) for _ in range(n)]
            if rep == 2 and n > 3:
                for _ in range(n // 3):
                    names[int(rng.integers(0, n))] = names[int(rng.integers(0, n))]
            got = H.name_order(names)
            assert got == R.hashbrown_name_order(names), (n, rep)
            assert sorted(names[i] for i in got) == sorted(set(names))
            last = {nm: i for i, nm in enumerate(names)}
            assert all(last[names[i]] == i for i in got)           # a repeated name carries its last sketch
    # nothing about the order is file order
    names = ["g%03d.fa" % i for i in range(64)]
    assert H.name_order(names) != list(range(64))


def test_gzip_reader_padding_fallback_and_zlib_only(tmp_path):
    """(i) bytes after the last member that do not start a member end the data — with 1 or many threads alike, and through
    the whole-file reader; (ii) when the fast decoder gives a member up midway (simulated), zlib takes it over from the
    member's start and the caller sees every byte exactly once; (iii) LASH_NO_FAST_INFLATE reads the same bytes with zlib."""
    import gzip
    import subprocess
    import sys
    rng = np.random.default_rng(21)
    parts = [bytes(np.frombuffer(b"ACGT\n", np.uint8)[rng.integers(0, 5, size=n)]) for n in (700_000, 5, 9_000_000, 123_456)]
    want = b"".join(parts)
    multi = b"".join(gzip.compress(p, 6) for p in parts)
    single = gzip.compress(want, 4)
    for name, data in (("multi", multi), ("single", single)):
        (tmp_path / (name + ".gz")).write_bytes(data)
        (tmp_path / (name + "_padded.gz")).write_bytes(data + b"\x00" * 1024)
    for name in ("multi", "single", "multi_padded", "single_padded"):
        for threads in (1, 4):
            got, _, _ = H.pgzip_read(str(tmp_path / (name + ".gz")), threads, 1 << 18)
            assert got == want, (name, threads)
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); import host_lib as H\n"
            "for name in ('multi', 'single'):\n"
            "    for threads in (1, 4):\n"
            "        for rs in (4097, 1 << 20):\n"
            "            d, p, s = H.pgzip_read(%r + '/' + name + '.gz', threads, rs)\n"
            "            assert d == open(%r, 'rb').read(), (name, threads, rs)\n"
            "print('same')\n" % (here, str(tmp_path), str(tmp_path / "want.bin")))
    (tmp_path / "want.bin").write_bytes(want)
    for env in ({"LASH_TEST_FAST_INFLATE_FAIL_AFTER": "0"}, {"LASH_TEST_FAST_INFLATE_FAIL_AFTER": "100000"},
                {"LASH_TEST_FAST_INFLATE_FAIL_AFTER": "5000000", "LASH_PGZIP_MEMBER_CAP": "1000000"}, {"LASH_NO_FAST_INFLATE": "1"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=os.path.dirname(here), **env))
        assert r.returncode == 0 and "same" in r.stdout, (env, r.stderr[-2000:])
    # (iv) a DEFECT in the fast decoder (simulated: one wrong output byte).  The sequential path hands bytes out before the
    # member's CRC is known; when zlib then takes the member over it re-makes those bytes, and their CRC must equal that of what
    # went out — a silent skip would leave the wrong byte with the caller.  The read must FAIL, not return other bytes.
    code2 = ("import sys; sys.path.insert(0, %r); import host_lib as H\n"
             "try:\n"
             "    d, p, s = H.pgzip_read(%r + '/single.gz', 1, 1 << 16)\n"
             "    print('returned', d == open(%r, 'rb').read())\n"
             "except ValueError as e:\n"
             "    print('refused:', e)\n" % (here, str(tmp_path), str(tmp_path / "want.bin")))
    for flip in ("12345", "6000000"):
        r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONPATH=os.path.dirname(here), LASH_TEST_FAST_INFLATE_FLIP_AT=flip))
        assert r.returncode == 0 and "refused:" in r.stdout and "disagree" in r.stdout, (flip, r.stdout, r.stderr[-2000:])


def test_name_order_with_odd_names():
    """names of every XXH3 length class (0 .. > 240 bytes once the 0xFF terminator is added), non-ASCII paths, one-letter and
    empty names: the two restatements agree and every name appears once"""
    pytest.importorskip("xxhash")
    import pyref as R
    names = ["", "a", "ab"] + ["n" * k for k in (3, 7, 8, 15, 16, 17, 127, 128, 129, 239, 240, 241, 300, 1025)]
    names += ["données/génome_%d.fa" % i for i in range(20)] + ["样本/基因组%d.fna.gz" % i for i in range(20)]
    got = H.name_order(names)
    assert got == R.hashbrown_name_order(names)
    assert sorted(got) == list(range(len(names)))


def test_block_formatter_equals_the_python_restatement(tmp_path):
    """host/dist_format.cpp (what `lash dist` and lash_amd.allpairs print rows with) against the pure-Python twin: list and matrix
    form, the triangle, "same name prints 0" (also for a name listed twice), 1 / 0 / NaN fast paths, several threads; the
    distances themselves come from lash_dist_rows on both sides (hmh statistics made with numpy)."""
    import lash_amd
    from lash_amd.allpairs import _Formatter, format_rows
    rng = np.random.default_rng(17)
    n, k = 23, 16
    names = ["g%d.fa" % i for i in range(n)]
    names[7] = names[3]                                            # listed twice
    names[11] = "dir with space/é.fa"
    card = rng.uniform(6e5, 5e6, n)
    c = rng.integers(0, 9000, size=(n, n)).astype(np.uint32)
    c[2, :] = 0                                                    # C == 0 -> similarity 0 -> distance exactly 1
    c[5, 4] = 16384
    nn = np.full((n, n), 16384, np.uint32)
    want = lash_amd.dist_rows("hmh", 0, k, 1, card, card, c_or_zero=c, n_counts=nn)
    for matrix in (False, True):
        f = _Formatter(names, card)
        path = tmp_path / ("m%d.txt" % matrix)
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        total = 0
        for b0, b1 in ((0, 1), (1, 9), (9, n)):                    # pair tables of a block: rows [b0, b1) x columns [0, b1)
            st = dict(c_or_zero=np.ascontiguousarray(c[b0:b1, :b1]), n_counts=np.ascontiguousarray(nn[b0:b1, :b1]))
            total += f.block(lash_amd.HMH, 0, k, 1, False, None, b0, b1, st, b1, matrix, 3, fd)
        os.close(fd)
        f.close()
        got = open(path, "rb").read().decode()
        assert len(got.encode()) == total
        assert got == format_rows(names, 0, want, matrix=matrix)
    assert "1.000000" in got and "0.000000" in got


def _fastq_records_after_errors(data, skip_bad):
    """independent restatement of the two hypotheses of layout.fastq_skip_bad, line-list based: every piece of
    data.split(b"\n") but the last is a newline-terminated line"""
    lines = data.split(b"\n")
    L = len(lines)
    recs, i = [], 0

    def record_at(i):
        if not lines[i].startswith(b"@") or i + 3 > L - 1:         # header, sequence and '+' lines must be terminated
            return None
        if not lines[i + 2].startswith(b"+"):
            return None
        s, q = lines[i + 1].rstrip(b"\r"), lines[i + 3].rstrip(b"\r")
        return s if len(s) == len(q) else None
    while i < L and not (i == L - 1 and lines[i] == b""):          # (the empty piece after a final newline is the end of the data)
        s = record_at(i)
        if s is not None:
            recs.append(s)
            i += 4
            continue
        if not skip_bad:
            break
        j = i + 1
        while j < L and not (lines[j].startswith(b"@") and j + 2 <= L - 1 and lines[j + 2].startswith(b"+")):
            j += 1
        i = j
    return recs


def test_fastq_error_switch_stop_or_skip():
    """layout.fastq_skip_bad (U6): utils.rs:457-458 keeps calling next() after an Err, so what lash sees after a malformed FASTQ
    record depends on needletail's iterator.  Both hypotheses — the iterator is finished (default) / the bad record is dropped and
    reading resumes at the next plausible record start — in the oracle and in lash_fastq_sanitize, against an independent
    restatement; the sanitized buffer parses to exactly the same records with EITHER rule (what the device parse needs)."""
    import ctypes as C
    import lash_amd
    import oracle_lib as O
    lib = lash_amd.load()
    rng = np.random.default_rng(23)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=40 + 3 * i)) for i in range(12)]
    rec = lambda i, q=None, plus=b"+": b"@r%d\n%s\n%s\n%s\n" % (i, seqs[i], plus, (b"I" * len(seqs[i])) if q is None else q)
    files = {
        "bad_plus_in_the_middle": rec(0) + rec(1) + rec(2) + rec(3, plus=b"-") + rec(4) + rec(5),
        "short_quality_in_the_middle": rec(0) + rec(1) + rec(6, q=b"III") + rec(7) + rec(8),
        "two_bad_stretches_and_a_bad_tail": rec(0) + b"garbage line\n" + rec(1) + rec(2, q=b"I") + rec(3) + b"@trunc\nACGT\n+\n",
        "bad_first_record": rec(9, plus=b"x") + rec(10) + rec(11),
        "well_formed": rec(0) + rec(1),
    }
    for name, data in files.items():
        for skip in (0, 1):
            want_recs = _fastq_records_after_errors(data, bool(skip))
            seq = np.frombuffer(b"".join(want_recs), np.uint8)
            off = np.concatenate([[0], np.cumsum([len(r) for r in want_recs])]).astype(np.uint64)
            want = O.sketch_genomes(O.HMH, 16, 0, 42, seq, off, np.array([0, len(want_recs)], np.uint64))[0]
            lay = O.make_layout(fastq_err="skip" if skip else "stop")
            got = O.sketch_files(O.HMH, 16, 0, 42, [data], layout=lay)[0]
            assert np.array_equal(got, want), (name, skip)
            buf = np.frombuffer(data, np.uint8).copy()
            changed = int(lib.lash_fastq_sanitize(buf.ctypes.data, len(data), skip))
            fixed = buf.tobytes()
            assert (changed == 0) == (name == "well_formed") and len(fixed) == len(data)
            for skip2 in (0, 1):                                  # now well-formed: both rules read the same records
                assert np.array_equal(O.sketch_files(O.HMH, 16, 0, 42, [fixed], layout=O.make_layout(fastq_err="skip" if skip2 else "stop"))[0], want), (name, skip, skip2)
    # the two rules differ exactly where records follow the error
    a = _fastq_records_after_errors(files["bad_plus_in_the_middle"], False)
    b = _fastq_records_after_errors(files["bad_plus_in_the_middle"], True)
    assert a == seqs[:3] and b == seqs[:3] + seqs[4:6]
