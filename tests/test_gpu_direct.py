"""GPU parity tests of the direct (ASCII-reading) sketch pass and its dirty-genome fallback (DESIGN.md "Direct mode").

lash_sketch_batch[_device] first lets the sketch kernel read the record bytes itself — exact while a genome holds only
upper-case ACGT, because filter_out_n (utils.rs:33-41) then deletes nothing — and re-does, in the same call, every genome
in which it met another byte through the pack stage.  Both routes must give the oracle's images, whatever the mix."""
import os
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O
import routes as R

pytestmark = pytest.mark.gpu
ALGO = {"hmh": 0, "hll": 1, "ull": 2}


@pytest.fixture(scope="module")
def ctx():
    import lash_amd
    c = lash_amd.Context(0)
    yield c
    c.close()


def oracle_images(an, k, p, seed, seq, off, goff):
    return O.sketch_genomes(ALGO[an], k, p, seed, seq, off, goff, threads=8)


def same(got, want, what):
    if not np.array_equal(got, want):
        rows = sorted({int(r) for r in np.argwhere(got != want)[:, 0]})
        raise AssertionError("%s: genomes %s differ" % (what, rows[:10]))


def clean_records(rng, n_rec, lo, hi):
    return ["".join(rng.choice("ACGT") for _ in range(rng.randint(lo, hi))).encode() for _ in range(n_rec)]


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hmh", 9, 0), ("hmh", 23, 0), ("hll", 21, 14), ("hll", 16, 8),
                                    ("ull", 16, 12), ("ull", 31, 10), ("ull", 4, 6)])
def test_clean_genomes_any_alignment(an, k, p):
    """Only ACGT: everything stays on the direct pass.  Odd lengths put every genome at a different byte alignment;
    multi-record genomes exercise the byte-position break bitmap; tiny genomes the byte-wise tail."""
    import lash_amd
    ctx = lash_amd.Context(0)               # fresh: no dirty-batch history that would make it pack first
    rng = random.Random(zlib.crc32(repr((an, k, p)).encode()))     # reproducible across processes (no PYTHONHASHSEED)
    gs = [clean_records(rng, 1, 1, 300) for _ in range(6)]
    gs += [clean_records(rng, rng.randint(2, 9), 0, 5000) for _ in range(8)]
    gs += [[O.synth_genome(700 + i, 100_000 + 37 * i + 1).tobytes()] for i in range(3)]
    gs += [[], [b""], [b"A"], [b"ACGT" * 24], [b"ACGT" * 23 + b"ACG"], [b"C" * 95, b"G" * 97]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    R.assert_ascii_route(tm)
    assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
    same(got, oracle_images(an, k, p, 42, seq, off, goff), "%s k=%d p=%d" % (an, k, p))
    same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), got, "NO_DIRECT")
    ctx.close()


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 12), ("ull", 19, 11)])
def test_mixed_clean_and_dirty_genomes(ctx, an, k, p):
    """A batch where some genomes are clean, some are dirty from the first byte, and some only near their end (so that
    most of their direct pass has already run when the flag goes up)."""
    import lash_amd
    rng = random.Random(5)
    L = 600_000
    gs = []
    for i in range(12):
        g = O.synth_genome(900 + i, L + i).copy()
        kind = i % 4
        if kind == 1:
            g[0] = ord("N")
        elif kind == 2:
            g[L - 5] = ord("a")                       # lower case is deleted too (utils.rs:36)
        elif kind == 3:
            g[L // 2: L // 2 + 1000] = ord("N")
        cuts = sorted({0, len(g)} | ({rng.randint(0, len(g)) for _ in range(3)} if i % 3 == 0 else set()))
        gs.append([g[a:b].tobytes() for a, b in zip(cuts[:-1], cuts[1:])])
    seq, off, goff = lash_amd.records_to_arrays(gs)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
    same(got, oracle_images(an, k, p, 42, seq, off, goff), "mixed")
    same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), got, "NO_DIRECT")


def test_every_byte_value_is_classified_like_filter_out_n(ctx):
    """One genome per byte value: ACGT flanks around a single foreign byte.  Only A, C, G, T may stay on the direct pass;
    the images must match the oracle for all 256."""
    import lash_amd
    flank = O.synth_genome(5, 500).tobytes()
    gs = [[flank + bytes([b]) + flank] for b in range(256)]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    got = ctx.sketch_batch("hmh", 11, 0, 42, seq, off, goff)
    same(got, oracle_images("hmh", 11, 0, 42, seq, off, goff), "byte values")


def test_accumulate_through_direct_pass(ctx):
    import lash_amd
    a = [[O.synth_genome(1, 40_000).tobytes()], [O.synth_genome(2, 50_001).tobytes()]]
    b = [[O.synth_genome(3, 30_003).tobytes()], [b"ACGTN" * 3000]]
    sa, oa, ga = lash_amd.records_to_arrays(a)
    sb, ob, gb = lash_amd.records_to_arrays(b)
    img = ctx.sketch_batch("ull", 16, 10, 42, sa, oa, ga)
    img = ctx.sketch_batch("ull", 16, 10, 42, sb, ob, gb, flags=lash_amd.F_ACCUMULATE, out=img)
    both = [[a[0][0], b[0][0]], [a[1][0], b[1][0]]]
    s2, o2, g2 = lash_amd.records_to_arrays(both)
    same(img, oracle_images("ull", 16, 10, 42, s2, o2, g2), "accumulate")


def test_direct_pass_backs_off_while_batches_are_dirty():
    """The context looks at the previous direct call's dirty-tile count (without waiting for it) and packs first while
    batches keep turning out dirty, probing again every 8th call.  Results never change."""
    import lash_amd
    c = lash_amd.Context(0)
    # finely fragmented dirt (every 97th byte lower-case: every wave-tile would be a compaction), so the direct pass hands these
    # genomes to the pack stage (sketch_kernels.hip: three dense wave-tiles in four)
    dirty = []
    for i in range(6):
        g = O.synth_genome(40 + i, 1_000_000).copy()
        g[::97] |= 0x20
        dirty.append([g.tobytes()])
    clean = [[O.synth_genome(50 + i, 1_000_000).tobytes()] for i in range(6)]
    sd, od, gd = lash_amd.records_to_arrays(dirty)
    sc, oc, gc = lash_amd.records_to_arrays(clean)
    want_d = oracle_images("hmh", 16, 0, 42, sd, od, gd)
    want_c = oracle_images("hmh", 16, 0, 42, sc, oc, gc)
    c.enable_timing(True)
    for _ in range(12):
        same(c.sketch_batch("hmh", 16, 0, 42, sd, od, gd), want_d, "dirty batch")      # synchronous: feedback has landed
    tried = c.timing()["direct_launches"]
    assert 2 <= tried <= 4, tried                      # call 1, then every 8th: far fewer than 12
    for _ in range(10):
        same(c.sketch_batch("hmh", 16, 0, 42, sc, oc, gc), want_c, "clean batch")
    assert c.timing()["direct_launches"] - tried >= 2  # back on once a probe saw a clean batch
    c.close()


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hmh", 7, 0), ("hll", 32, 10), ("ull", 17, 9)])
def test_every_short_length_on_the_direct_pass(an, k, p):
    """Clean genomes of every length 0..420 in one batch (so every byte alignment and every position of the genome end
    relative to a lane's 96-byte window occurs), once as single records and once cut into two records."""
    import lash_amd
    ctx = lash_amd.Context(0)
    rng = random.Random(11)
    base = O.synth_genome(123, 1000).tobytes()
    gs = []
    for n in range(421):
        s = base[n % 7: n % 7 + n]
        gs.append([s])
        cut = rng.randint(0, n)
        gs.append([s[:cut], s[cut:]])
    seq, off, goff = lash_amd.records_to_arrays(gs)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    R.assert_ascii_route(tm)
    assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
    same(got, oracle_images(an, k, p, 42, seq, off, goff), "short lengths %s k=%d" % (an, k))
    ctx.close()


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 10), ("ull", 13, 9)])
def test_many_small_genomes_write_their_own_images(an, k, p):
    """A batch of thousands of small genomes: each is a single work item that writes its image (header included) from the
    sketch kernel; only a census kernel follows.  Clean and dirty genomes, genomes shorter than k (empty image owed),
    union into existing images, and the pack-first route must all agree with the oracle."""
    import lash_amd
    ctx = lash_amd.Context(0)
    rng = random.Random(17)
    base = O.synth_genome(321, 400_000).tobytes()
    gs = []
    for i in range(3000):
        n = rng.choice([rng.randint(0, 40), rng.randint(40, 3000), rng.randint(3000, 12_000)])
        s = bytearray(base[(i * 97) % 300_000:][:n])
        if n and rng.random() < 0.3:
            j = rng.randrange(n)
            s[j:j + rng.choice([1, 5, 200])] = b"N" * len(s[j:j + rng.choice([1, 5, 200])])
        cut = rng.randint(0, n) if rng.random() < 0.2 else n
        gs.append([bytes(s[:cut]), bytes(s[cut:])] if cut < n else [bytes(s)])
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(an, k, p, 42, seq, off, goff)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    if R.sole_on():
        assert tm["sole_launches"] == 1 and tm["direct_launches"] == 0, tm                   # whole genomes on persistent workgroups, no work items at all
    else:
        assert tm["sketch_workgroups"] == sum(1 for g in gs if sum(len(r) for r in g) > 0)     # one work item per non-empty genome
    assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
    same(got, want, "small genomes " + an)
    same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, "pack-first")
    # union of two halves of every genome's records into the same images
    s1, o1, g1 = lash_amd.records_to_arrays([g[:1] for g in gs])
    s2, o2, g2 = lash_amd.records_to_arrays([g[1:] for g in gs])
    acc = ctx.sketch_batch(an, k, p, 42, s1, o1, g1)
    acc = ctx.sketch_batch(an, k, p, 42, s2, o2, g2, flags=lash_amd.F_ACCUMULATE, out=acc)
    same(acc, want, "accumulate")
    ctx.close()


def _sparse_dirt_genomes(rng, k):
    """genomes with few, short deletions at positions chosen to hit lane (64 B), wave (4 KiB), tile and slice boundaries,
    the genome's ends, record boundaries, and each other (runs closer than k)"""
    gs, L = [], 300_000
    for i in range(10):
        g = bytearray(O.synth_genome(1200 + i, L + 13 * i).tobytes())
        spots = [0, 1, k - 1, k, 63, 64, 65, 4095, 4096, 4097, 32767, 32768, len(g) - 1, len(g) - k, len(g) - k - 1,
                 len(g) // 2, 100_000 + i, 100_000 + i + k - 2, 100_000 + i + k + 1]
        for s_ in rng.sample(spots, 6):
            run = rng.choice([1, 1, 2, k - 1, k, k + 1, 37, 100, 129, 700])
            fill = rng.choice([b"N", b"n", b"a", b"R", b"\x00", b"\xff", b"-"])
            if rng.random() < 0.3:
                fill = bytes(rng.choice(b"NnacgtRYKMSW") for _ in range(run))
                g[s_:s_ + run] = fill[:max(0, min(run, len(g) - s_))]
            else:
                g[s_:s_ + run] = fill * max(0, min(run, len(g) - s_))
        cuts = {0, len(g)}
        if i % 2:
            cuts |= {rng.randint(0, len(g)) for _ in range(4)} | {100_000 + i + 3, 4096, 64}
        cuts = sorted(cuts)
        gs.append([bytes(g[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    # a short genome that is mostly dirt, one that is only dirt, one with a single valid k-mer spread over three islands
    gs.append([b"ACGT" * 5 + b"N" * 50 + b"ACGTTGCA" * 6])
    gs.append([b"N" * 500])
    island = b"ACGTTGCATGCATCGATCGGATTACAGGATC"
    gs.append([island[:5] + b"nn" + island[5:11] + b"N" * 90 + island[11:]])
    return gs


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hmh", 11, 0), ("hmh", 31, 0), ("hmh", 1, 0), ("hll", 21, 14), ("hll", 32, 9),
                                    ("ull", 16, 12), ("ull", 14, 16), ("ull", 5, 4)])
def test_sparse_dirt_is_handled_in_place(an, k, p):
    """Few short deletions per genome: the direct pass keeps the genome (no pack-stage fallback: pack time stays ~0) and
    hashes the junction k-mers — those whose window spans a deleted byte — from the joined flanks, exactly as
    filter_out_n + the per-record iterator do (utils.rs:33-41, 457-499).  Census and surviving-base count included."""
    import lash_amd
    ctx = lash_amd.Context(0)
    rng = random.Random(zlib.crc32(repr(("sparse", an, k, p)).encode()))
    gs = _sparse_dirt_genomes(rng, k)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(an, k, p, 42, seq, off, goff)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    R.assert_ascii_route(tm)
    same(got, want, "sparse dirt in place %s k=%d" % (an, k))
    assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
    assert tm["bases_last"] == sum(len(O.filter_out_n(r)) for g in gs for r in g)
    same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, "pack-first")
    ctx.close()


def test_long_runs_and_dense_dirt_still_fall_back(ctx):
    """A long gap (compacted in place by its wave: dense_tile) and a genome with soft-masked confetti all over (handed to
    stream_sketch_kernel in the same call): images, census and surviving-base count as the oracle's."""
    import lash_amd
    a = bytearray(O.synth_genome(1300, 400_000).tobytes())
    a[200_000:230_000] = b"N" * 30_000                                  # one long gap
    b = np.frombuffer(O.synth_genome(1301, 400_000).tobytes(), np.uint8).copy()
    b[::53] |= 0x20                                                      # soft-masked confetti
    c = bytearray(O.synth_genome(1302, 400_000).tobytes())
    c[123_456] = ord("N")                                               # stays in place
    gs = [[bytes(a)], [b.tobytes()], [bytes(c)]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    for an, k, p in (("hmh", 16, 0), ("hll", 25, 12), ("ull", 16, 10)):
        ctx.enable_timing(True)
        got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
        tm = ctx.timing()
        ctx.enable_timing(False)
        same(got, oracle_images(an, k, p, 42, seq, off, goff), "fallback " + an)
        assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
        assert tm["bases_last"] == sum(len(O.filter_out_n(r)) for g in gs for r in g)


def _soft_masked_genomes(rng):
    """records with lower-case / N stretches of 50 bytes .. 300 kb at random places: whole wave-tiles deleted, tiles and wave parts that
    end inside a run, runs across record and slice boundaries, a gap longer than the stream kernel's 8 KiB skip"""
    gs = []
    for n_rec in (1, 1, 3, 17):
        recs = []
        for _ in range(n_rec):
            L = rng.choice([5_000, 70_000, 400_000, 1_500_000]) if n_rec < 17 else rng.randint(0, 30_000)
            s = bytearray(O.synth_genome(rng.randint(0, 10**6), max(L, 1)).tobytes()[:L])
            pos = rng.randrange(max(L, 1))
            while pos < L:
                run = min(rng.choice([50, 300, 2_000, 4_096, 10_000, 45_000, 300_000]), L - pos)
                if rng.random() < 0.5:
                    s[pos:pos + run] = bytes(s[pos:pos + run]).lower()
                else:
                    s[pos:pos + run] = b"N" * run
                pos += run + rng.choice([1, 15, 16, 31, 33, 500, 2_048, 10_000, 100_000])
            recs.append(bytes(s))
        gs.append(recs)
    return gs


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hmh", 5, 0), ("hmh", 32, 0), ("hll", 21, 14), ("hll", 16, 16), ("ull", 16, 12), ("ull", 27, 17)])
def test_stream_kernel_alone_equals_the_oracle(an, k, p):
    """LASH_F_STREAM_ONLY: every genome through stream_sketch_kernel (what the context does by itself while batches keep turning
    out soft-masked) — clean, sparsely dirty, soft-masked, multi-record, every short length; census and surviving bases too."""
    import lash_amd
    ctx = lash_amd.Context(0)
    rng = random.Random(zlib.crc32(repr(("stream", an, k, p)).encode()))
    corpora = [_soft_masked_genomes(rng), _sparse_dirt_genomes(rng, k),
               [clean_records(rng, 1, 100_000, 600_000), clean_records(rng, 40, 0, 9_000), [b"ACGT" * 7, b"", b"N" * 5000, b"acgt" * 3000]],
               [[O.synth_genome(9000 + L, max(L, 1)).tobytes()[:L]] for L in list(range(0, 70)) + [2047, 2048, 2049, 4095, 4096, 4097, 16383, 16385]]]
    for gs in corpora:
        seq, off, goff = lash_amd.records_to_arrays(gs)
        want = oracle_images(an, k, p, 42, seq, off, goff)
        ctx.enable_timing(True)
        got = ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_STREAM_ONLY)
        tm = ctx.timing()
        ctx.enable_timing(False)
        assert tm["direct_launches"] == 0
        same(got, want, "stream only %s k=%d" % (an, k))
        assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
        assert tm["bases_last"] == sum(len(O.filter_out_n(r)) for g in gs for r in g)
        # and through the default route (direct pass, in-place compaction, hand-over) and the pack-first one
        same(ctx.sketch_batch(an, k, p, 42, seq, off, goff), want, "default route")
        same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, "pack first")
    ctx.close()


def test_a_gap_longer_than_the_look_ahead_scan(ctx):
    """An assembly gap of 5 MB (centromere-sized): the wave of the direct pass that reaches it gives up after scanning 4 MiB
    (DENSE_SCAN_MAX) and hands the genome over; the stream kernel skips the gap 8 KiB per round trip.  The k-mers that span the
    gap are the ones filter_out_n's joined flanks give."""
    import lash_amd
    g = bytearray(O.synth_genome(4242, 9_000_000).tobytes())
    g[2_000_123:7_100_456] = b"N" * (7_100_456 - 2_000_123)
    h = bytearray(O.synth_genome(4243, 6_000_000).tobytes())
    h[100:5_600_000] = bytes(h[100:5_600_000]).lower()                   # a soft-masked genome with a clean tail
    gs = [[bytes(g)], [bytes(h)], [O.synth_genome(4244, 300_000).tobytes()]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    for an, k, p in (("hmh", 16, 0), ("ull", 31, 12)):
        want = oracle_images(an, k, p, 42, seq, off, goff)
        ctx.enable_timing(True)
        got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
        tm = ctx.timing()
        ctx.enable_timing(False)
        same(got, want, "long gap " + an)
        assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g_ in gs for r in g_)
        assert tm["bases_last"] == sum(len(O.filter_out_n(r)) for g_ in gs for r in g_)
        same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_STREAM_ONLY), want, "long gap, stream only " + an)


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("ull", 21, 12)])
def test_many_unequal_genomes_are_handed_out_longest_first(an, k, p):
    """More work items than workgroup slots and sizes that differ: the library orders the launch longest first (item_order) and
    the kernels index partials, censuses and images through that permutation.  700 genomes of 20 kb .. 600 kb, a few of them
    soft-masked (handed to the stream kernel under the same order), some in several records."""
    import lash_amd
    ctx = lash_amd.Context(0)
    rng = random.Random(zlib.crc32(repr(("order", an, k, p)).encode()))
    gs = []
    for i in range(700):
        L = int(20_000 * (30 ** rng.random()))
        s = bytearray(O.synth_genome(7000 + i, L).tobytes())
        if i % 37 == 5:
            for b in range(0, L, 3000):
                s[b:b + 700] = bytes(s[b:b + 700]).lower()
        if i % 11 == 3:
            cuts = sorted(rng.sample(range(1, L), 3))
            gs.append([bytes(s[a:b]) for a, b in zip([0] + cuts, cuts + [L])])
        else:
            gs.append([bytes(s)])
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(an, k, p, 42, seq, off, goff)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    same(got, want, "unequal genomes " + an)
    assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g)
    assert tm["bases_last"] == sum(len(O.filter_out_n(r)) for g in gs for r in g)
    same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_STREAM_ONLY), want, "unequal genomes, stream only " + an)
    same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, "unequal genomes, pack first " + an)
    ctx.close()


@pytest.mark.parametrize("frac", [0.002, 0.05])
def test_a_read_set_with_ns_is_voted_on_by_its_waves(ctx, frac):
    """One 30 Mbp genome of 150-bp reads, an N in `frac` of them: hundreds of work items, so the hand-over needs 1 in 32 of the
    genome's waves to agree (GenomeDesc::handover) — a few unlucky waves at 0.2 % do not send it away, at 5 % they all vote.
    Either way the images, the census and the surviving bases are the oracle's."""
    import lash_amd
    rng = np.random.default_rng(int(frac * 1e4))
    n, rl = 200_000, 150
    s = np.frombuffer(O.synth_genome(5151, n * rl).tobytes(), np.uint8).copy()
    hit = rng.choice(n, size=int(n * frac), replace=False)
    s[hit * rl + rng.integers(0, rl, size=len(hit))] = ord("N")
    seq = s
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(rl))
    goff = np.array([0, n], dtype=np.uint64)
    for an, k, p in (("ull", 16, 12), ("hmh", 21, 0)):
        want = oracle_images(an, k, p, 42, seq, off, goff)
        ctx.enable_timing(True)
        got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
        tm = ctx.timing()
        ctx.enable_timing(False)
        same(got, want, "reads with N %s %g" % (an, frac))
        assert tm["kmers"] == (n - len(hit)) * (rl - k + 1) + len(hit) * (rl - 1 - k + 1)      # (the N is deleted, its flanks are joined)
        assert tm["bases_last"] == n * rl - len(hit)


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("ull", 21, 12), ("hll", 21, 16)])
def test_the_last_round_of_equal_items_is_cut_into_quarters(an, k, p):
    """More equal work items than workgroup slots: the last round's worth of them is cut into quarters (lash_api.hip, "the tail of
    a launch").  220 genomes of 1.06 Mbp are 1 100 slices of 13 252 words; with 512 slots (hmh on a 256-CU part) the first 588 stay
    whole and the other 512 become four work items each — genomes with 5, 8..17 and 20 partials in one launch, which also takes
    the finalize stage through its grouped fold.  Some genomes in records, one soft-masked; the oracle checks a sample from both ends and
    the middle, the census all of it.  hll p = 16 runs two bucket-space passes per slice on top."""
    import lash_amd
    ctx = lash_amd.Context(0)
    n, L = 220, 1_060_000
    gs = []
    for i in range(n):
        s = O.synth_genome(9000 + i, L).tobytes()
        if i % 53 == 1:
            gs.append([s[:300_001], s[300_001:300_050], s[300_050:]])
        elif i == 120:
            b = bytearray(s)
            for at in range(0, L, 5000):
                b[at:at + 900] = bytes(b[at:at + 900]).lower()
            gs.append([bytes(b)])
        else:
            gs.append([s])
    seq, off, goff = lash_amd.records_to_arrays(gs)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    assert tm["kmers"] == sum(max(0, len(O.filter_out_n(r)) - k + 1) for g in gs for r in g)
    import torch
    if (an == "hmh" and torch.cuda.get_device_properties(0).multi_processor_count == 256 and "LASH_TAIL_SPLIT" not in os.environ
            and "LASH_SLICE_FACTOR" not in os.environ and "LASH_TAIL_GEO" not in os.environ):
        # round 4: batches of equal genomes whose HyperMinHash launch may defer signatures are cut once per slot, not twice, and
        # their tail into halves: 220 x 3 slices, the last 512 of them in two (ull / hll keep five slices and quarters: 1 100 -> 2 636);
        # round 6: the later half of that last round in four, its last quarter in eight (sizes cannot see what a byte costs)
        assert tm["sketch_workgroups"] == (660 - 512) + 256 * 2 + 128 * 4 + 128 * 8   # (64 KiB of LDS sketch: two workgroups per CU)
    assert tm["sketch_workgroups"] > (660 if an == "hmh" else 1100)
    sample = [0, 1, 2, 14, 15, 16, 54, 107, 116, 117, 118, 119, 120, 121, 160, n - 2, n - 1]
    sseq, soff, sgoff = lash_amd.records_to_arrays([gs[i] for i in sample])
    same(got[sample], oracle_images(an, k, p, 42, sseq, soff, sgoff), "tail quarters " + an)
    ctx.close()
