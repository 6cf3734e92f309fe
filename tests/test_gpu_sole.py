"""GPU parity tests of the persistent small-genome kernel (lash_amd/csrc/sole_kernels.hip, round 5; run with -m gpu).

Reference semantics: one sketch per file whatever its size (/root/reference/src/utils.rs:450-509), filter_out_n deletes every byte
that is not upper-case ACGT and joins the flanks (utils.rs:33-41), k-mers never span records (utils.rs:457-499).  Every test calls
through the C ABI, checks that the new kernel really ran (lash_timing.sole_launches) and compares images, the k-mer census and the
surviving-base count with the CPU oracle."""
import os
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_parity import ALGO, messy_genomes

pytestmark = [pytest.mark.gpu, pytest.mark.sole]


@pytest.fixture(autouse=True, params=[None, "1", "3"], ids=["wgs=default", "wgs=1", "wgs=3"])
def _few_workgroups(request, monkeypatch):
    """Every test three times: with the launch the library plans (a test batch of a hundred genomes then has a workgroup per genome), and
    with ONE and THREE workgroups (LASH_SOLE_WGS), so that many genomes follow each other on the same rings and table with the next
    one's bytes in flight — what a collection of a million looks like to a workgroup.  (Round 5: the first hash pass of a genome wiped
    its early record starts when the genome before it on the same workgroup had needed more than one pass — `zero_w` kept its last
    value; nothing in the suite put two such genomes on one workgroup, tests/test_gpu_fullsize.py::test_a_viral_collection_at_full_size found it.)"""
    if request.param is None:
        monkeypatch.delenv("LASH_SOLE_WGS", raising=False)
    else:
        monkeypatch.setenv("LASH_SOLE_WGS", request.param)
    yield


@pytest.fixture(scope="module")
def ctx():
    import lash_amd
    c = lash_amd.Context(0)
    yield c
    c.close()


def oracle_images(an, k, p, seed, seq, off, goff, x_low=0, layout=None):
    return O.sketch_genomes(ALGO[an], k, p, seed, seq, off, goff, threads=8, hmh_x_is_low=x_low, layout=layout)


def same(got, want, what):
    if not np.array_equal(got, want):
        rows = sorted({int(r) for r in np.argwhere(got != want)[:, 0]})
        raise AssertionError("%s: genomes %s differ" % (what, rows[:12]))


def run(ctx, an, k, p, gs, seed=42, flags=0, expect_sole=True, what=""):
    """sketch the genomes, check launch counters and the two censuses, return the images"""
    import lash_amd
    seq, off, goff = lash_amd.records_to_arrays(gs)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, seed, seq, off, goff, flags=flags)
    tm = ctx.timing()
    ctx.enable_timing(False)
    if expect_sole is not None:
        assert (tm["sole_launches"] >= 1) == expect_sole, (what, tm)
    assert tm["kmers"] == sum(len(O.record_kmers(r, k)) for g in gs for r in g), what
    assert tm["bases_last"] == sum(len(O.filter_out_n(r)) for g in gs for r in g), what
    return got, (seq, off, goff), tm


PARAMS = [("hmh", 16, 0), ("hmh", 8, 0), ("hmh", 15, 0), ("hmh", 17, 0), ("hmh", 32, 0), ("hmh", 21, 0),
          ("hll", 21, 14), ("hll", 16, 10), ("hll", 5, 4), ("hll", 32, 12), ("hll", 21, 11), ("hll", 13, 13),
          ("ull", 16, 12), ("ull", 1, 3), ("ull", 19, 13), ("ull", 27, 8), ("ull", 32, 10), ("ull", 21, 4)]


@pytest.mark.parametrize("an,k,p", PARAMS)
def test_messy_batches(ctx, an, k, p):
    """Ns, IUPAC codes, lower case, empty records, records shorter than k, genomes without records: every route agrees with the oracle."""
    import lash_amd
    rng = random.Random(zlib.crc32(repr(("sole", an, k, p)).encode()))
    gs = messy_genomes(rng, 41)
    gs[5] = []
    gs[7] = [b"", b"", b"ACGT"]
    gs[9] = [b"N" * 70, b"acgt" * 9, b""]
    gs[11] = [b"ACGTACGTAC"] * 40                                    # many records shorter than k (for k > 10)
    got, (seq, off, goff), tm = run(ctx, an, k, p, gs, what="direct")
    assert tm["direct_launches"] == 0                                # nothing but small genomes: the persistent kernel alone
    want = oracle_images(an, k, p, 42, seq, off, goff)
    same(got, want, "sole %s k=%d p=%d" % (an, k, p))
    got2, _, _ = run(ctx, an, k, p, gs, flags=lash_amd.F_NO_DIRECT, what="packed")      # the pack stage's 2-bit stream as the source
    same(got2, want, "sole packed")
    got3, _, _ = run(ctx, an, k, p, gs, flags=lash_amd.F_NO_SOLE, expect_sole=False, what="no sole")
    same(got3, want, "sliced")


@pytest.mark.parametrize("k", list(range(1, 33)))
def test_every_k(ctx, k):
    rng = random.Random(770 + k)
    gs = messy_genomes(rng, 9, max_rec=4, max_len=1500)
    for an, p in (("hmh", 0), ("ull", 9), ("hll", 8)):
        got, (seq, off, goff), _ = run(ctx, an, k, p, gs, seed=1234567)
        same(got, oracle_images(an, k, p, 1234567, seq, off, goff), "%s k=%d" % (an, k))


@pytest.mark.parametrize("threads", [64, 128, 256, 512])
@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 10), ("ull", 31, 9)])
def test_lengths_around_round_boundaries(ctx, threads, an, k, p, monkeypatch):
    """A round is 16 bytes per lane: genomes that end just before / at / after 1, 2 and 3 rounds of every workgroup shape, clean, with a
    deleted run across the boundary, and cut into records at and around it."""
    monkeypatch.setenv("LASH_SOLE_THREADS", str(threads))
    R = 16 * threads
    base = O.synth_genome(321, 3 * R + 400).tobytes()
    rng = random.Random(threads)
    gs = []
    for m in (1, 2, 3):
        for d in (-33, -17, -16, -15, -1, 0, 1, 15, 16, 17, 31, 32, 47):
            n = m * R + d
            s = base[d % 5: d % 5 + n]
            gs.append([s])
            gs.append([s[:m * R - 3] + b"N" * 7 + s[m * R + 4:]])                       # deleted bytes across the round boundary
            gs.append([s[:m * R - 40] + b"n" * 90 + s[m * R + 50:]])                    # ... a longer run
            gs.append([s[:m * R], s[m * R:]])                                           # a record start exactly on it
            c = m * R + rng.randint(-20, 20)
            gs.append([s[:c - 5], s[c - 5:c], s[c:]])                                   # short record next to it
            gs.append([s[:c] + b"NNN", b"NN" + s[c:]])                                  # record start inside a deleted run
    got, (seq, off, goff), _ = run(ctx, an, k, p, gs)
    same(got, oracle_images(an, k, p, 42, seq, off, goff), "round boundaries T=%d" % threads)


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 10), ("ull", 13, 9), ("hmh", 25, 0)])
def test_every_short_length(ctx, an, k, p):
    """Clean genomes of every length 0..420 in one batch (every byte alignment of a genome's first byte), once as single records and
    once cut in two; then the same with one deleted byte somewhere."""
    rng = random.Random(11)
    base = O.synth_genome(123, 1000).tobytes()
    gs = []
    for n in range(421):
        s = base[n % 7: n % 7 + n]
        gs.append([s])
        cut = rng.randint(0, n)
        gs.append([s[:cut], s[cut:]])
        if n:
            j = rng.randrange(n)
            gs.append([s[:j] + b"N" + s[j + 1:]])
    got, (seq, off, goff), _ = run(ctx, an, k, p, gs)
    same(got, oracle_images(an, k, p, 42, seq, off, goff), "short lengths")


def test_every_byte_value(ctx):
    flank = O.synth_genome(5, 500).tobytes()
    gs = [[flank + bytes([b]) + flank] for b in range(256)]
    for an, k, p in (("hmh", 16, 0), ("hll", 21, 12)):
        got, (seq, off, goff), _ = run(ctx, an, k, p, gs)
        same(got, oracle_images(an, k, p, 42, seq, off, goff), "byte values")


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 12), ("ull", 19, 11)])
def test_deleted_runs_everywhere(ctx, an, k, p):
    """Runs of 1..700 deleted bytes at lane (16), wave (1 024) and round boundaries, closer than k to each other, at both ends."""
    rng = random.Random(3)
    L = 30000
    gs = []
    for i in range(60):
        g = bytearray(O.synth_genome(400 + i, L).tobytes())
        for _ in range(rng.randint(1, 8)):
            run_len = rng.choice([1, 2, 15, 16, 17, 31, 100, 700])
            at = rng.choice([0, 16, 1024, 8192, 16384, L - run_len, rng.randrange(L)]) + rng.choice([-1, 0, 1])
            at = max(0, min(L - run_len, at))
            g[at:at + run_len] = bytes(rng.choice(b"NnacgtRY-") for _ in range(run_len))
        cuts = sorted({0, L} | {rng.randrange(L) for _ in range(rng.randint(0, 3))})
        gs.append([bytes(g[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    gs.append([b"N" * 20000])
    gs.append([b"acgt" * 5000 + b"ACGTTGCA" * 10])
    gs.append([b"N" * 9000 + O.synth_genome(1, 200).tobytes() + b"N" * 9000])
    got, (seq, off, goff), _ = run(ctx, an, k, p, gs)
    same(got, oracle_images(an, k, p, 42, seq, off, goff), "deleted runs")


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 10), ("ull", 13, 9)])
def test_thousands_of_small_genomes_and_accumulate(ctx, an, k, p):
    """Many chunks on every workgroup: 6 000 genomes of 0..3 000 bases, a tenth dirty, some multi-record; then a second batch of the same
    size united into the same images (LASH_F_ACCUMULATE) against the oracle's merge."""
    import lash_amd
    rng = random.Random(17)
    pool = O.synth_genome(9, 400000).tobytes()

    def batch(salt):
        gs = []
        for i in range(6000):
            n = rng.choice([0, 5, rng.randint(1, 300), rng.randint(300, 3000)])
            a = rng.randrange(len(pool) - n)
            s = bytearray(pool[a:a + n])
            if n and rng.random() < 0.1:
                s[rng.randrange(n)] = ord(rng.choice("NnRacgt"))
            s = bytes(s)
            if n > 10 and rng.random() < 0.2:
                c = rng.randrange(n)
                gs.append([s[:c], s[c:]])
            else:
                gs.append([s])
        return gs
    a, b = batch(0), batch(1)
    got_a, (sa, oa, ga), tm = run(ctx, an, k, p, a)
    assert tm["direct_launches"] == 0
    want_a = oracle_images(an, k, p, 42, sa, oa, ga)
    same(got_a, want_a, "first batch")
    sb, ob, gb = lash_amd.records_to_arrays(b)
    got_ab = ctx.sketch_batch(an, k, p, 42, sb, ob, gb, flags=lash_amd.F_ACCUMULATE, out=got_a.copy())
    want_b = oracle_images(an, k, p, 42, sb, ob, gb)
    want_ab = np.stack([O.merge_images(ALGO[an], p, want_a[i], want_b[i]) for i in range(len(a))])
    same(got_ab, want_ab, "accumulate")
    got_ab2 = ctx.sketch_batch(an, k, p, 42, sb, ob, gb, flags=lash_amd.F_ACCUMULATE | lash_amd.F_NO_DIRECT, out=got_a.copy())
    same(got_ab2, want_ab, "accumulate, packed")


@pytest.mark.sole(3000)
@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 12), ("ull", 16, 10)])
def test_batches_with_both_kinds_of_genomes(ctx, an, k, p):
    """LASH_SOLE_MAX=3000: genomes below go to the persistent kernel, the others are cut into work items, in one call; finalize leaves the
    small ones alone, the censuses count every genome once."""
    import lash_amd
    rng = random.Random(23)
    gs = messy_genomes(rng, 60, max_rec=5, max_len=2500)              # 0 .. 12 500 bytes per genome
    gs += [[O.synth_genome(50 + i, 40000 + i).tobytes()] for i in range(3)]
    n_small = sum(1 for g in gs if sum(len(r) for r in g) <= 3000)
    assert 5 < n_small < len(gs) - 5
    got, (seq, off, goff), tm = run(ctx, an, k, p, gs)
    assert tm["direct_launches"] == 1
    want = oracle_images(an, k, p, 42, seq, off, goff)
    same(got, want, "mixed, direct")
    got2, _, _ = run(ctx, an, k, p, gs, flags=lash_amd.F_NO_DIRECT)
    same(got2, want, "mixed, packed")
    acc = ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_ACCUMULATE, out=want.copy())
    same(acc, want, "mixed, union with itself")


def test_layouts_and_x_low(ctx):
    """The image writers of the persistent kernel honour the context's layout (headers, big-endian HyperMinHash registers) and
    LASH_F_HMH_X_LOW."""
    import lash_amd
    rng = random.Random(29)
    gs = messy_genomes(rng, 17, max_rec=3, max_len=4000)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    got, _, _ = run(ctx, "hmh", 16, 0, gs, flags=lash_amd.F_HMH_X_LOW)
    same(got, oracle_images("hmh", 16, 0, 42, seq, off, goff, x_low=1), "x low")
    for spec, an, k, p in (("hmh_reg=be,hmh_hdr=lp", "hmh", 16, 0), ("hll_hdr=pZsa", "hll", 21, 10), ("ull_hdr=Lp", "ull", 16, 9),
                           ("codes=TGCA", "hmh", 18, 0), ("hmh_x=low", "hmh", 12, 0)):
        try:
            ctx.set_layout(spec)
            got, _, _ = run(ctx, an, k, p, gs)
            want = oracle_images(an, k, p, 42, seq, off, goff, layout=O.parse_layout(spec))
            same(got, want, spec)
        finally:
            ctx.set_layout(None)


def test_device_entry_and_raw_files(ctx):
    """lash_sketch_batch_device on resident bytes, and FASTA / FASTQ file bytes (device parse -> pack stage -> persistent kernel)."""
    import torch
    import lash_amd
    rng = random.Random(31)
    gs = messy_genomes(rng, 50, max_rec=4, max_len=5000)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    gbo = off[goff.astype(np.int64)]
    d_seq = torch.from_numpy(seq.copy()).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    for an, k, p in (("hmh", 16, 0), ("hll", 21, 10)):
        ib = lash_amd.image_bytes(an, p)
        d_img = torch.zeros(len(gs) * ib, dtype=torch.uint8, device="cuda")
        ctx.enable_timing(True)
        ctx.sketch_batch_device(an, k, p, 42, d_seq, d_off, len(off) - 1, goff, gbo, d_img)
        ctx.synchronize()
        tm = ctx.timing()
        ctx.enable_timing(False)
        assert tm["sole_launches"] == 1 and tm["direct_launches"] == 0
        want = oracle_images(an, k, p, 42, seq, off, goff)
        same(d_img.cpu().numpy().reshape(len(gs), ib), want, "device entry")
        files = []
        for g in gs:
            if rng.random() < 0.5:
                files.append(b"".join(b">r\n" + b"\n".join(r[i:i + 60] for i in range(0, len(r), 60)) + b"\n" for r in g) or b">empty\n")
            else:
                files.append(b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in g) or b"@e\n\n+\n\n")
        ctx.enable_timing(True)
        got = ctx.sketch_files_raw(an, k, p, 42, files)
        tm = ctx.timing()
        ctx.enable_timing(False)
        assert tm["sole_launches"] >= 1
        same(got, O.sketch_files(ALGO[an], k, p, 42, files, threads=8), "raw files")


@pytest.mark.parametrize("an,k,p", [("hll", 16, 13), ("hll", 21, 14), ("hmh", 16, 0), ("ull", 16, 12)])
def test_genomes_without_a_kmer_between_others(ctx, an, k, p):
    """Two image flushes follow each other with no hash pass between them when a genome has no k-mer at all (an empty file, a file of
    records shorter than k, the empty half of an accumulating call).  Round 5: the HyperLogLog flush zeroed its histogram on the first
    wave while the other waves were already tallying the next genome's registers — `sum` lost counts (registers right, three header
    bytes off), only with several genomes on one workgroup (the fixture's wgs=1 / wgs=3) and a table large enough for several waves."""
    rng = random.Random(5)
    base = O.synth_genome(77, 60000).tobytes()
    gs = []
    for i in range(60):
        r = i % 6
        if r == 0: gs.append([base[rng.randint(0, 1000):][:rng.randint(2000, 40000)]])
        elif r == 1: gs.append([b""])
        elif r == 2: gs.append([base[:k - 1], base[5:5 + k - 1]])                      # records shorter than k
        elif r == 3: gs.append([b"N" * 50])
        elif r == 4: gs.append([base[100:100 + k]])                                    # exactly one k-mer
        else: gs.append([b"", base[300:300 + rng.randint(k, 3000)], b"acgtn"])
    got, (seq, off, goff), _ = run(ctx, an, k, p, gs)
    want = oracle_images(an, k, p, 42, seq, off, goff)
    same(got, want, "genomes without k-mers")
    # the same as the second half of an accumulating call: every genome's image holds its first record already
    import lash_amd
    first = [[g[0]] for g in gs]
    rest = [g[1:] if len(g) > 1 else [b""] for g in gs]
    s1, o1, g1 = lash_amd.records_to_arrays(first)
    s2, o2, g2 = lash_amd.records_to_arrays(rest)
    img = ctx.sketch_batch(an, k, p, 42, s1, o1, g1)
    img = ctx.sketch_batch(an, k, p, 42, s2, o2, g2, flags=lash_amd.F_ACCUMULATE, out=img)
    same(img, want, "accumulate over genomes without k-mers")


def test_bases_last_counts_small_genomes_when_the_context_streams_first():
    """once a context has learnt that its batches are soft-masked (`stream_first`: the optimistic direct pass is skipped, every
    sliced genome's flag is up), lash_timing.bases_last subtracts the stream kernel's deleted-byte counts — and the persistent kernel must
    write the small genomes' counts THERE too.  A batch of large soft-masked genomes teaches the context, then a mixed batch whose small
    genomes hold N runs is sketched: images, census and surviving bases against the oracle."""
    import lash_amd
    c = lash_amd.Context(0)
    large = []
    for i in range(6):
        g = O.synth_genome(940 + i, 1_000_000).copy()
        g[::97] |= 0x20
        large.append([g.tobytes()])
    sd, od, gd = lash_amd.records_to_arrays(large)
    for _ in range(3):
        c.sketch_batch("hmh", 16, 0, 42, sd, od, gd)
    small = []
    for i in range(9):
        s = bytearray(O.synth_genome(960 + i, 20_000 + 777 * i).tobytes())
        s[5_000 + i:5_300 + i] = b"N" * 300
        s[100:140] = bytes(s[100:140]).lower()
        small.append([bytes(s)])
    gs = small[:4] + large[:2] + small[4:] + large[2:4]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    c.enable_timing(True)
    got = c.sketch_batch("hmh", 16, 0, 42, seq, off, goff)
    tm = c.timing()
    c.enable_timing(False)
    assert tm["sole_launches"] == 1 and tm["direct_launches"] == 0, tm          # the context streamed first: no optimistic pass in this call
    assert tm["bases_last"] == sum(len(O.filter_out_n(r)) for g in gs for r in g), tm
    assert tm["kmers"] == sum(len(O.record_kmers(r, 16)) for g in gs for r in g)
    same(got, oracle_images("hmh", 16, 0, 42, seq, off, goff), "mixed batch on a context that streams first")
    c.close()


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 12), ("ull", 16, 10)])
def test_a_record_that_belongs_to_no_genome_is_not_sketched(ctx, an, k, p):
    """genome_rec_off = [0, 1, 1] with two records is legal input — genome 1 is empty and record 1 belongs to nobody.  It has as many
    records as genomes and no multi-record genome, which is what the all-small route took for "the record offsets ARE the genome byte
    offsets"; it then sketched the stray record as genome 1, past the declared end of the sequence bytes.  Also with the stray record in front."""
    a, b = O.synth_genome(970, 9_000).tobytes(), O.synth_genome(971, 7_000).tobytes()
    for goff, off, seq, keep in (([0, 1, 1], [0, 9_000, 16_000], a + b, a), ([0, 0, 1], [0, 9_000, 16_000], a + b, None)):
        seq = np.frombuffer(seq, np.uint8)
        off, goff = np.array(off, np.uint64), np.array(goff, np.uint64)
        if keep is None:
            # genome 0 empty, genome 1 = record 0, record 1 belongs to nobody
            pass
        want = oracle_images(an, k, p, 42, seq, off, goff)
        ctx.enable_timing(True)
        got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
        tm = ctx.timing()
        ctx.enable_timing(False)
        same(got, want, "stray record %s" % goff.tolist())
        empty = oracle_images(an, k, p, 42, np.zeros(1, np.uint8), np.array([0, 0], np.uint64), np.array([0, 1], np.uint64))[0]
        assert np.array_equal(got[0 if keep is None else 1], empty)              # the empty genome's image is the empty sketch
        assert tm["kmers"] == 9_000 - k + 1, tm
