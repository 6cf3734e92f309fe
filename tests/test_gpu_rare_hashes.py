"""k-mers whose hash has 32 and more leading zeros above / below the bucket bits: one in 2^32 and rarer, i.e. never met by random test data,
yet each kernel family has code for exactly them —
  * the word loops take the rank from ONE 32-bit `v_ffbh` and re-run a word whose rank bits were all zero with the exact 64-bit form
    (sketch_rules.h `z_redo`);
  * UltraLogLog tables in LDS keep a 64-bit nlz bitmap per register as two words: nlz >= 32 lands in the second;
  * round 6: `bins_apply_kernel` (UltraLogLog p = 18 .. 22) keeps only the bitmap's LOW word in LDS; an entry with nlz >= 32 goes to the genome's
    fallback table in global memory and the registers are read out of both;
  * HyperLogLog ranks above 32 (and, above 53 - p, the `sum` corner with its replay: tests/test_gpu_hll_corner.py pins four found k-mers).
xxh3_64 of 8 bytes is a bijection (tests/pyref.py `xxh3_64_8b_inverse`), so the test BUILDS such k-mers: choose the hash, invert it, keep the
value if it is a canonical k-mer (k = 32: every 64-bit value is a 32-mer; k = 31 / 28: the top bits must be zero too), spell it in ACGT.
Bit-exact against the oracle on every route, like every other parity test."""
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu

ALGO = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}
SEED = 42


@pytest.fixture(scope="module")
def ctx():
    import lash_amd
    c = lash_amd.Context(0)
    yield c
    c.close()


def _spell(v, k):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k)).encode()


def _kmer_with_hash(make_hash, k, rng, tries=20000):
    """A canonical k-mer (as ASCII) whose xxh3_64 is make_hash(rng): the free bits of the hash are redrawn until the preimage fits 2k bits and is
    its own canonical form."""
    for _ in range(tries):
        h = make_hash(rng)
        v = R.xxh3_64_8b_inverse(h, SEED)
        if k < 32 and v >> (2 * k):
            continue
        km = _spell(v, k)
        if O.record_kmers(km, k)[0] == v:
            assert O.xxh3_64_8b(v, SEED) == h
            return km, h
    raise AssertionError("no canonical preimage found")


# fewer free bits in the hash than this: the register is drawn too (a preimage is a canonical k-mer once in 2 / 2^3 / 2^9 draws)
FREE_BITS = {32: 5, 31: 8, 28: 14}


def _ull_hash(p, idx, nlz, k):
    """hash4j's split: idx = h >> (64 - p), nlz = leading zeros of the 64 - p bits below it."""
    q = 64 - p
    def make(rng):
        if nlz >= q:
            return rng.getrandbits(p) << q
        below = q - 1 - nlz                                    # bits under the leading one
        i = idx if below >= FREE_BITS[k] else rng.getrandbits(p)
        return (i << q) | (1 << below) | rng.getrandbits(below)
    return make


def _hll_hash(p, bucket, rho, k):
    """streaming_algorithms' split: bucket = h & (2^p - 1), rho = (64 - p) - bitlen(h >> p) + 1."""
    bl = 64 - p - rho + 1                                      # bit length of h >> p
    def make(rng):
        if bl == 0:
            return rng.getrandbits(p)
        b = bucket if bl - 1 >= FREE_BITS[k] else rng.getrandbits(p)
        return ((1 << (bl - 1)) | rng.getrandbits(bl - 1)) << p | b
    return make


def _genomes(kms, rng):
    """The built k-mers as records of their own between random records (no neighbours), inline in one long record (neighbouring k-mers overlap
    them), and in a genome of several work items."""
    rnd = O.synth_genome(rng.randrange(1 << 20), 2_600_000).tobytes()
    a, at = [], 0
    for km in kms:
        a += [rnd[at:at + 3000], km]; at += 3000
    inline = b"".join(rnd[100_000 + 5000 * i:100_000 + 5000 * (i + 1)] + km for i, km in enumerate(kms))
    big = rnd[200_000:1_400_000] + kms[0] + b"N" + kms[-1] + rnd[1_400_000:2_600_000]
    return [a, [inline], [big], [kms[len(kms) // 2]]]


@pytest.mark.parametrize("k", [32, 31, 28])
@pytest.mark.parametrize("p", [10, 12, 14, 16, 17, 18, 20, 22, 23, 24])
def test_ultraloglog_kmers_with_32_and_more_leading_zeros(ctx, p, k):
    import lash_amd
    rng = random.Random(zlib.crc32(repr(("ull", p, k)).encode()))
    q = 64 - p
    top = (1 << p) - 1
    # (register, nlz): alone in their registers; 31 + 32 and 32 + 33 + 34 sharing one (the two bits below the top come from both words of the
    # bitmap); the first and the last register of the table and of a bin; the largest nlz there is.  (Where the hash has too few free bits
    # left the register is drawn, not chosen: _ull_hash.)
    plan = [(5, 32), (6, 33), (7, min(40, q - 1)), (top, q), (0, 35), (1 << (p - 1), q - 1), (top, 33),
            (9, 31), (9, 32), (11, 32), (11, 33), (11, 34), ((1 << 14) - 1 if p > 14 else 3, 36), (1 << 14 if p > 14 else 4, 32)]
    kms, held = [], {}
    for idx, nlz in plan:
        km, h = _kmer_with_hash(_ull_hash(p, idx & top, min(nlz, q), k), k, rng)
        kms.append(km)
        held.setdefault(h >> q, set()).add(min(nlz, q))
    gs = _genomes(kms, rng)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(ALGO["ull"], k, p, seq, off, goff)
    hdr = want.shape[1] - (1 << p)
    # the oracle agrees on what these registers hold: 4 * (nlz + p - 1) of the rarest k-mer, and the bits of the two ranks below it
    for idx, zs in held.items():
        z = max(zs)
        assert int(want[0, hdr + idx]) == ((z + p - 1) << 2) | (2 if z - 1 in zs else 0) | (1 if z - 2 in zs else 0), (idx, zs)
    assert len(held) >= 8 and (k != 32 or any(len(zs) == 3 for zs in held.values()))
    got = ctx.sketch_batch("ull", k, p, SEED, seq, off, goff)
    assert np.array_equal(got, want), "ull p=%d k=%d: %d bytes differ" % (p, k, int((got != want).sum()))
    got = ctx.sketch_batch("ull", k, p, SEED, seq, off, goff, flags=lash_amd.F_NO_DIRECT)
    assert np.array_equal(got, want), "ull p=%d k=%d pack-first: %d bytes differ" % (p, k, int((got != want).sum()))


@pytest.mark.parametrize("p", [18, 20, 22, 23])
def test_more_rare_entries_in_one_bin_than_its_short_list_holds(ctx, p):
    """bins_apply_kernel keeps the entries with nlz >= 32 of one (genome, bin) in a list of 62 beside its table; the 63rd and later go to the
    genome's fallback table in global memory (and the workgroup then pays an agent-scope fence).  Hashed input never gets there; 150 built k-mers
    in one quarter of the table's first 2^16 registers (one bin, whichever size bins have) do — some sharing a register with each other and with the list's entries."""
    import lash_amd
    k = 32
    rng = random.Random(p)
    q = 64 - p
    kms, held = [], {}
    for i in range(150):
        idx = (3 << 14) | rng.choice((rng.randrange(1 << 14), 7, 8, (1 << 14) - 1))       # bin 3 of 2^14 registers = the upper half of bin 1 of 2^15
        nlz = rng.choice((31, 32, 33, 34, 35))
        km, h = _kmer_with_hash(_ull_hash(p, idx, nlz, k), k, rng)
        assert h >> q == idx
        kms.append(km)
        held.setdefault(idx, set()).add(nlz)
    rnd = O.synth_genome(4242 + p, 900_000).tobytes()
    gs = [[rnd[:400_000]] + kms + [rnd[400_000:]], [b"".join(kms)], [rnd[:50_000]]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(ALGO["ull"], k, p, seq, off, goff)
    hdr = want.shape[1] - (1 << p)
    for idx, zs in held.items():
        z = max(zs)
        assert int(want[0, hdr + idx]) == ((z + p - 1) << 2) | (2 if z - 1 in zs else 0) | (1 if z - 2 in zs else 0), (idx, zs)
    for flags in (0, lash_amd.F_NO_DIRECT):
        got = ctx.sketch_batch("ull", k, p, SEED, seq, off, goff, flags=flags)
        assert np.array_equal(got, want), "ull p=%d flags=%d: %d bytes differ" % (p, flags, int((got != want).sum()))
    # ... and the fallback table was left empty: the next call on the same context (same buffers) sees none of it
    seq2, off2, goff2 = lash_amd.records_to_arrays([[rnd[:300_000]], [rnd[300_000:700_000]]])
    assert np.array_equal(ctx.sketch_batch("ull", k, p, SEED, seq2, off2, goff2), oracle_images(ALGO["ull"], k, p, seq2, off2, goff2))


@pytest.mark.parametrize("k", [32, 28])
@pytest.mark.parametrize("p", [10, 14, 16])
def test_hyperloglog_ranks_above_32(ctx, p, k):
    import lash_amd
    rng = random.Random(zlib.crc32(repr(("hll", p, k)).encode()))
    top = (1 << p) - 1
    exact = 53 - p                                             # up to here `sum` is order-free (above: the corner and its replay)
    plan = [(5, 32), (6, 33), (7, 34), (0, min(37, exact)), (top, exact), (9, 31), (9, 33), (12, exact + 1), (13, 65 - p)]
    kms, held = [], {}
    for b, rho in plan:
        km, h = _kmer_with_hash(_hll_hash(p, b, rho, k), k, rng)
        kms.append(km)
        held[h & top] = max(held.get(h & top, 0), rho)
    gs = _genomes(kms, rng)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(ALGO["hll"], k, p, seq, off, goff)
    hdr = want.shape[1] - (1 << p)
    for b, rho in held.items():
        assert int(want[0, hdr + b]) == rho, (b, rho)
    assert len(held) >= 7 and max(held.values()) == 65 - p
    got = ctx.sketch_batch("hll", k, p, SEED, seq, off, goff)                     # (the host entry replays `sum` where a rank exceeds 53 - p)
    assert np.array_equal(got, want), "hll p=%d k=%d: %d bytes differ" % (p, k, int((got != want).sum()))
    got = ctx.sketch_batch("hll", k, p, SEED, seq, off, goff, flags=lash_amd.F_NO_DIRECT)
    assert np.array_equal(got, want), "hll p=%d k=%d pack-first: %d bytes differ" % (p, k, int((got != want).sum()))


def oracle_images(algo, k, p, seq, off, goff):
    return O.sketch_genomes(algo, k, p, SEED, seq, off, goff, threads=8)
