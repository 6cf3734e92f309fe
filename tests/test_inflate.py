"""lash_amd/csrc/host/inflate_fast.cpp (DEFLATE decoder, CRC-32, gzip framing, bounded-memory reader) against zlib:
every block type, compression level and strategy, code lengths beyond the first-level tables, short-period and far
matches, multi-member files, optional header fields, truncation at every byte, random corruption.  CPU only."""
import gzip
import io
import random
import struct
import zlib

import numpy as np
import pytest

import host_lib as H


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return c.compress(data) + c.flush()


def _corpus():
    rng = np.random.default_rng(7)
    dna = rng.choice(np.frombuffer(b"ACGT", np.uint8), 400_000).tobytes()
    fasta = b">chr1 test\n" + b"\n".join(dna[i:i + 80] for i in range(0, len(dna), 80)) + b"\n"
    qual = rng.integers(33, 74, 60_000, dtype=np.uint8).tobytes()
    fastq = b"".join(b"@read%d/1\n" % i + dna[i * 150:(i + 1) * 150] + b"\n+\n" + qual[i * 150:(i + 1) * 150] + b"\n" for i in range(400))
    skew = bytes(min(255, int(x)) for x in rng.geometric(0.08, 300_000))          # long-tailed alphabet: code lengths up to 15
    periodic = b"".join((bytes([65 + i]) * (i + 1)) * 3000 for i in range(9)) + b"N" * 100_000 + b"ACGTTGCA" * 20_000
    return {
        "empty": b"", "one": b"A", "short": b"hello hello hello", "dna": dna, "fasta": fasta, "fastq": fastq, "skew": skew,
        "periodic": periodic, "random": rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(),
        "soft_masked": bytes(c | 0x20 if (i // 5000) % 2 else c for i, c in enumerate(dna[:200_000])),
    }


CORPUS = _corpus()


@pytest.mark.parametrize("name", sorted(CORPUS))
def test_every_level_and_strategy_matches_zlib(name):
    data = CORPUS[name]
    for level in (0, 1, 4, 6, 9):
        assert H.gunzip(_gz(data, level)) == data, (name, level)
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        assert H.gunzip(_gz(data, 6, strategy)) == data, (name, strategy)
    assert H.gunzip(_gz(data, 9, mem=1)) == data                       # tiny hash table: many small blocks


def test_crc32_matches_zlib():
    rng = random.Random(3)
    blob = bytes(rng.randrange(256) for _ in range(70_000))
    for n in list(range(0, 300)) + [1000, 4095, 4096, 4097, 65_535, 70_000]:
        for off in (0, 1, 3, 7):
            d = blob[off:off + n]
            assert H.crc32(d) == zlib.crc32(d), (n, off)
            assert H.crc32(d, 0xDEADBEEF) == zlib.crc32(d, 0xDEADBEEF)
    assert H.crc32(blob[30_000:], H.crc32(blob[:30_000])) == zlib.crc32(blob)   # incremental


def test_multi_member_headers_and_padding():
    a, b, c = CORPUS["fasta"][:50_000], CORPUS["fastq"][:30_000], b""
    buf = io.BytesIO()
    with gzip.GzipFile(filename="some name.fa", mode="wb", fileobj=buf, mtime=12345) as g:          # FNAME
        g.write(a)
    m1 = buf.getvalue()
    m2 = _gz(b)
    # FEXTRA + FCOMMENT + FHCRC by hand around a raw deflate body
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(a[:1000]) + raw.flush()
    hdr = bytes([0x1f, 0x8b, 8, 4 | 16 | 2, 0, 0, 0, 0, 0, 255]) + struct.pack("<H", 6) + b"BC\x02\x00\x12\x34" + b"a comment\x00"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    m3 = hdr + body + struct.pack("<II", zlib.crc32(a[:1000]), 1000)
    m4 = _gz(c)                                                          # an empty member
    want = a + b + a[:1000] + c
    assert H.gunzip(m1 + m2 + m3 + m4) == want
    assert H.gunzip(m1 + m2 + m3 + m4 + b"\x00" * 512) == want           # tar-style zero padding is ignored, as zlib's gzread does
    assert H.gunzip(m1 + b"garbage that is not a member") == a
    for bad in (b"", b"\x1f\x8b", b"not gzip at all, really not", m1[:10]):
        with pytest.raises(ValueError):
            H.gunzip(bad)
    with pytest.raises(ValueError):
        H.gunzip(bytes([0x1f, 0x8b, 7]) + m1[3:])                         # unknown method


@pytest.mark.parametrize("name", ["fasta", "fastq", "periodic", "skew", "random", "one", "empty"])
def test_bounded_memory_reader(name):
    data = CORPUS[name]
    for level, window, piece in ((6, 300, 1), (6, 1000, 777), (1, 4096, 100_000), (0, 500, 64), (9, 1 << 20, 1 << 16)):
        blob = _gz(data, level) + _gz(b"second member")
        got, used = H.gunzip_windowed(blob, window, piece)
        assert got == data and used == len(_gz(data, level)), (name, level, window, piece)


def test_truncation_and_corruption_never_pass():
    data = CORPUS["fastq"][:20_000]
    blob = _gz(data, 6)
    for cut in list(range(0, 40)) + list(range(len(blob) - 40, len(blob))) + [len(blob) // 3, len(blob) // 2]:
        with pytest.raises(ValueError):
            H.gunzip(blob[:cut])
        with pytest.raises(ValueError):
            H.gunzip_windowed(blob[:cut], 1000, 100)
    rng = random.Random(11)
    survived = 0
    for trial in range(1500):
        b = bytearray(blob)
        for _ in range(rng.choice([1, 1, 2, 5])):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        try:
            got = H.gunzip(bytes(b))
        except ValueError:
            continue
        survived += 1
        # only flips in header fields nobody checks (MTIME, XFL, OS) leave the stream valid
        assert got == data, trial
    assert survived < 200


def test_streams_zlib_rejects_are_rejected():
    """distance before the start of the stream, invalid block type, stored-length mismatch, a code set that is
    over-subscribed, a dynamic block without an end-of-block code: raw deflate bodies built bit by bit"""
    def member(body, payload=b""):
        return bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255]) + body + struct.pack("<II", zlib.crc32(payload), len(payload))

    class Bits:
        def __init__(self):
            self.v, self.n = 0, 0
        def put(self, val, n):
            self.v |= val << self.n
            self.n += n
        def huff(self, code, n):                       # Huffman codes go MSB first
            for i in range(n - 1, -1, -1):
                self.put((code >> i) & 1, 1)
        def bytes(self):
            return self.v.to_bytes((self.n + 7) // 8, "little")

    b = Bits(); b.put(1, 1); b.put(3, 2)                                       # block type 3
    with pytest.raises(ValueError):
        H.gunzip(member(b.bytes()))
    b = Bits(); b.put(1, 1); b.put(0, 2); b.put(0, 5); b.put(5, 16); b.put(5, 16)   # LEN == NLEN
    with pytest.raises(ValueError):
        H.gunzip(member(b.bytes() + b"hello"))
    # fixed block: length symbol 257 (code 0000001, 7 bits) + distance code 0 (5 bits) at the very start: distance 1 > 0 bytes
    b = Bits(); b.put(1, 1); b.put(1, 2); b.huff(0b0000001, 7); b.huff(0, 5); b.huff(0, 7)
    with pytest.raises(ValueError):
        H.gunzip(member(b.bytes()))
    # fixed block: literal 'A' (0x30 + 65 = 8-bit code), then length 3 distance 1 -> "AAAA"; valid
    b = Bits(); b.put(1, 1); b.put(1, 2); b.huff(0x30 + 65, 8); b.huff(0b0000001, 7); b.huff(0, 5); b.huff(0, 7)
    assert H.gunzip(member(b.bytes(), b"AAAA")) == b"AAAA"
    assert zlib.decompress(member(b.bytes(), b"AAAA"), 31) == b"AAAA"
    # fixed block using litlen symbol 286 (8-bit code 11000110): invalid
    b = Bits(); b.put(1, 1); b.put(1, 2); b.huff(0b11000110, 8)
    with pytest.raises(ValueError):
        H.gunzip(member(b.bytes()))
    # dynamic block whose code-length code is over-subscribed (three codes of length 1)
    b = Bits(); b.put(1, 1); b.put(2, 2); b.put(0, 5); b.put(0, 5); b.put(0, 4)
    for _ in range(3):
        b.put(1, 3)
    b.put(0, 3)
    with pytest.raises(ValueError):
        H.gunzip(member(b.bytes() + b"\x00" * 8))
