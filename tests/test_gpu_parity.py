"""GPU parity tests (run on the MI355X box with -m gpu): liblash_gfx950.so, called through its C ABI, must give
bit-identical sketch images to the CPU oracle on the same inputs."""
import hashlib
import json
import os
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O
import routes as R
from fastx import read_fastx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGO = {"hmh": 0, "hll": 1, "ull": 2}


@pytest.fixture(scope="module")
def ctx():
    import lash_amd
    c = lash_amd.Context(0)
    yield c
    c.close()


def oracle_images(algo, k, p, seed, seq, rec_off, goff, x_low=0):
    return O.sketch_genomes(algo, k, p, seed, seq, rec_off, goff, threads=8, hmh_x_is_low=x_low)


def assert_same(got, want, what=""):
    assert got.shape == want.shape, what
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError("%s: %d bytes differ, first at %s (got %d want %d)" % (
            what, len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))


def messy_genomes(rng, n_genomes, max_rec=6, max_len=3000):
    alphabet = "ACGT" * 12 + "NnacgtRYKM-"
    gs = []
    for _ in range(n_genomes):
        recs = []
        for _ in range(rng.randint(0, max_rec)):
            mode = rng.random()
            n = rng.randint(0, max_len)
            if mode < 0.5:
                s = "".join(rng.choice("ACGT") for _ in range(n))
            elif mode < 0.8:
                s = "".join(rng.choice(alphabet) for _ in range(n))
            elif mode < 0.9:
                s = "N" * n
            else:
                s = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 40)))
            recs.append(s.encode())
        gs.append(recs)
    return gs


def test_fixture_files_match_oracle_and_digests(ctx):
    import lash_amd
    want_dig = json.load(open(os.path.join(GOLD, "fixture_digests.json")))
    for key, dig in sorted(want_dig.items()):
        name, an, k, p, _ = key.split("|")
        k, p = int(k[1:]), int(p[1:])
        recs = read_fastx(os.path.join(GOLD, name))
        seq, off, goff = lash_amd.records_to_arrays([recs])
        img = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
        assert hashlib.sha256(img[0].tobytes()).hexdigest() == dig, key


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hmh", 8, 0), ("hmh", 15, 0), ("hmh", 17, 0), ("hmh", 32, 0),
                                    ("hll", 21, 14), ("hll", 16, 10), ("hll", 5, 4), ("hll", 31, 15), ("hll", 32, 12),
                                    ("ull", 16, 12), ("ull", 1, 3), ("ull", 19, 14), ("ull", 27, 8), ("ull", 32, 10)])
def test_messy_batches_match_oracle(ctx, an, k, p):
    import lash_amd
    rng = random.Random(zlib.crc32(repr((an, k, p)).encode()))     # reproducible across processes (no PYTHONHASHSEED)
    gs = messy_genomes(rng, 23)
    gs[5] = []                              # genome without records
    gs[7] = [b"", b"", b"ACGT"]             # empty records
    seq, off, goff = lash_amd.records_to_arrays(gs)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    want = oracle_images(ALGO[an], k, p, 42, seq, off, goff)
    assert_same(got, want, "%s k=%d p=%d" % (an, k, p))


@pytest.mark.parametrize("k", list(range(1, 33)))
def test_every_k_hmh_and_ull(ctx, k):
    import lash_amd
    rng = random.Random(77 + k)
    gs = messy_genomes(rng, 6, max_rec=4, max_len=1500)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    for an, p in (("hmh", 0), ("ull", 9), ("hll", 8)):
        got = ctx.sketch_batch(an, k, p, 1234567, seq, off, goff)
        want = oracle_images(ALGO[an], k, p, 1234567, seq, off, goff)
        assert_same(got, want, "%s k=%d" % (an, k))


@pytest.mark.parametrize("seed", [0, 42, 93, 2**63 + 5, 2**64 - 1])
def test_seeds(ctx, seed):
    import lash_amd
    gs = [[O.synth_genome(g, 30000).tobytes()] for g in range(3)]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    for an, k, p in (("hmh", 16, 0), ("hll", 21, 12), ("ull", 16, 12)):
        assert_same(ctx.sketch_batch(an, k, p, seed, seq, off, goff),
                    oracle_images(ALGO[an], k, p, seed, seq, off, goff), "%s seed=%d" % (an, seed))


def test_hmh_x_low_switch(ctx):
    import lash_amd
    gs = [[O.synth_genome(9, 50000).tobytes()]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    got = ctx.sketch_batch("hmh", 16, 0, 42, seq, off, goff, flags=lash_amd.F_HMH_X_LOW)
    assert_same(got, oracle_images(0, 16, 0, 42, seq, off, goff, x_low=1), "x_low")
    assert not np.array_equal(got, ctx.sketch_batch("hmh", 16, 0, 42, seq, off, goff))


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hll", 21, 14), ("ull", 16, 12), ("hmh", 31, 0), ("ull", 11, 14)])
def test_multi_slice_genomes(ctx, an, k, p):
    """Genomes long enough to be cut into several workgroup slices and many lane tiles; N runs and record
    boundaries placed near slice/tile/word boundaries."""
    import lash_amd
    rng = np.random.default_rng(11)
    g0 = O.synth_genome(100, 3_000_000)
    g1 = O.synth_genome(101, 1_234_567).copy()
    g1[500_000:500_777] = ord("N")
    g1[1_000_001] = ord("n")
    cuts = [0, 15, 16, 17, 31, 32, 33, 4095, 4096, 4097, 32768, 32769, 262144 - 1, 262144, 262144 + 1, 700_000, 1_234_567]
    recs1 = [g1[a:b].tobytes() for a, b in zip(cuts[:-1], cuts[1:])]
    g2 = O.synth_genome(102, 70_001)
    seq, off, goff = lash_amd.records_to_arrays([[g0.tobytes()], recs1, [g2.tobytes()]])
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    want = oracle_images(ALGO[an], k, p, 42, seq, off, goff)
    assert_same(got, want, "%s k=%d p=%d" % (an, k, p))


def test_short_reads_fastq_like(ctx):
    """cfg5 shape: thousands of 150-bp reads, some with N, into one sketch."""
    import lash_amd
    rng = random.Random(5)
    reads = []
    for i in range(4000):
        r = [rng.choice("ACGT") for _ in range(150)]
        if i % 17 == 0:
            r[rng.randrange(150)] = "N"
        if i % 501 == 0:
            r = r[:12]
        reads.append("".join(r).encode())
    seq, off, goff = lash_amd.records_to_arrays([reads])
    for an, k, p in (("ull", 16, 12), ("hmh", 16, 0), ("hll", 21, 10), ("ull", 31, 12)):
        assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff), oracle_images(ALGO[an], k, p, 42, seq, off, goff), an)


@pytest.mark.parametrize("an,k,p", [("hll", 16, 16), ("hll", 25, 16), ("hll", 21, 16), ("ull", 16, 15), ("ull", 12, 16), ("ull", 21, 17), ("ull", 21, 16), ("ull", 32, 15),
                                    ("ull", 21, 18), ("ull", 9, 19), ("ull", 16, 20), ("ull", 30, 21), ("ull", 16, 22), ("ull", 16, 23), ("ull", 21, 24)])
def test_register_tables_larger_than_lds(ctx, an, k, p):
    """2^p registers beyond 128 KiB of LDS as 32-bit words.  Round 4: hll p=16 and ull p=15..17 keep them as BYTES (one pass, updates
    behind a filter: LdsByteQRegs); ull p=18..23 are BINNED — every k-mer hashed once, an entry appended to the list of its bin
    (2^15 registers), one LDS pass per bin (bins_apply_kernel); ull p >= 24 keeps its table in HBM/L2 and takes one global atomic per k-mer.  Multi-record, multi-slice, dirty and empty genomes, through the direct route and the
    pack-first route."""
    import lash_amd
    g = O.synth_genome(8, 1_300_000)
    gs = [[O.synth_genome(7, 200_000).tobytes()], [b"ACGTNACGT" * 50], [], [g[:700_000].tobytes(), g[700_000:].tobytes()]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(ALGO[an], k, p, 42, seq, off, goff)
    ctx.enable_timing(True)
    assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff), want, an)
    assert ctx.timing()["kmers"] == sum(len(O.record_kmers(r, k)) for g_ in gs for r in g_)
    ctx.enable_timing(False)
    assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, an + " pack-first")


@pytest.mark.parametrize("an,k,p", [("hll", 16, 16), ("hll", 21, 16), ("ull", 16, 15), ("ull", 21, 16), ("ull", 11, 17)])
def test_byte_tables_behind_their_filter_on_reads(ctx, an, k, p):
    """hll p = 16 / ull p = 15 .. 17 keep their registers as bytes in LDS; the word loop asks each register's byte whether the k-mer would
    change it and stacks those that would (LdsByteQRegs), the exact compare-and-swap update runs from the stacks.  Reads make every tile
    a masked one (record boundaries), equal lengths take the computed boundaries, unequal ones the bitmap; an N per few reads walks
    junctions beside the stacks; enough reads that a lane's stack fills many times over."""
    import lash_amd
    rng = random.Random(zlib.crc32(repr((an, k, p)).encode()))
    body = O.synth_genome(k + p, 2_400_000).tobytes()
    equal = [body[i:i + 150] for i in range(0, 1_200_000, 150)]
    ragged, at = [], 1_200_000
    while at < 2_400_000:
        n = rng.choice((150, 150, 149, 151, 75, 250, k, k - 1 if k > 1 else 1))
        r = bytearray(body[at:at + n])
        if rng.random() < 0.05 and len(r) > 2:
            r[rng.randrange(len(r))] = ord("N")
        ragged.append(bytes(r)); at += n
    gs = [equal, ragged, [body[:300_000]]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(ALGO[an], k, p, 42, seq, off, goff)
    ctx.enable_timing(True)
    assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff), want, an + " reads")
    assert ctx.timing()["kmers"] == sum(len(O.record_kmers(r, k)) for g_ in gs for r in g_)
    ctx.enable_timing(False)
    assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, an + " reads, pack-first")


@pytest.mark.parametrize("an,k,p", [("ull", 16, 18), ("ull", 21, 20), ("ull", 16, 22), ("ull", 21, 23), ("hll", 21, 16), ("ull", 12, 16)])
def test_large_tables_on_repeats_take_the_fallback_paths(ctx, an, k, p):
    """Binned tables (ull p >= 18) stage entries in per-bin rows and append them to per-bin lists sized for hashed — i.e. spread — k-mers.
    A satellite-like genome puts millions of identical k-mers into a handful of buckets: rows overflow (the word is re-run straight into
    the fallback table) and lists overflow (the rest is spilled there, bins_apply_kernel reads it beside its table).  Byte tables (hll p = 16, ull
    p = 16) see the same input as compare-and-swap contention on few words.  All of it must still be the oracle's image."""
    import lash_amd
    rnd = O.synth_genome(77, 400_000).tobytes()
    unit = b"ACGTTGCATTAGC"
    gs = [[(unit * 120_000)[:1_500_000]],                                   # one tandem repeat: 13 distinct canonical k-mer phases
          [rnd[:150_000] + b"A" * 900_000 + rnd[150_000:300_000]],          # a homopolymer run between unique flanks
          [rnd[:200_000], (b"AC" * 400_000), b"ACGT"],                      # records; a dinucleotide repeat
          [rnd]]                                                            # an ordinary genome beside them (same group, its own lists)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    want = oracle_images(ALGO[an], k, p, 42, seq, off, goff)
    ctx.enable_timing(True)
    assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff), want, an + " repeats")
    assert ctx.timing()["kmers"] == sum(len(O.record_kmers(r, k)) for g_ in gs for r in g_)
    ctx.enable_timing(False)
    assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, an + " repeats, pack-first")


def test_parameter_errors_mirror_reference_panics(ctx):
    import lash_amd
    seq, off, goff = lash_amd.records_to_arrays([[b"ACGT" * 10]])
    for an, k, p in (("hmh", 0, 0), ("hmh", 33, 0), ("hll", 16, 3), ("hll", 16, 17), ("ull", 16, 2), ("ull", 16, 27)):
        with pytest.raises(lash_amd.LashError) as e:
            ctx.sketch_batch(an, k, p, 42, seq, off, goff)
        assert e.value.code == lash_amd.EINVAL
    with pytest.raises(lash_amd.LashError):
        ctx.sketch_batch("minhash", 16, 0, 42, seq, off, goff)


def test_empty_batch_and_empty_genomes(ctx):
    import lash_amd
    seq, off, goff = lash_amd.records_to_arrays([])
    assert ctx.sketch_batch("hmh", 16, 0, 42, seq, off, goff).shape == (0, 32768)
    seq, off, goff = lash_amd.records_to_arrays([[], [b""], [b"NNNN"], [b"ACG"]])
    for an, k, p in (("hmh", 16, 0), ("hll", 16, 10), ("ull", 16, 10)):
        assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff), oracle_images(ALGO[an], k, p, 42, seq, off, goff), an)


def test_merge_and_accumulate(ctx):
    import lash_amd
    a = [[O.synth_genome(g, 40000).tobytes()] for g in range(4)]
    b = [[O.synth_genome(50 + g, 25000).tobytes()] for g in range(4)]
    sa, oa, ga = lash_amd.records_to_arrays(a)
    sb, ob, gb = lash_amd.records_to_arrays(b)
    both = [x + y for x, y in zip(a, b)]
    sc, oc, gc = lash_amd.records_to_arrays(both)
    for an, k, p in (("hmh", 16, 0), ("hll", 21, 14), ("ull", 16, 12), ("ull", 16, 16)):
        ia = ctx.sketch_batch(an, k, p, 42, sa, oa, ga)
        ib = ctx.sketch_batch(an, k, p, 42, sb, ob, gb)
        whole = oracle_images(ALGO[an], k, p, 42, sc, oc, gc)
        merged = ctx.merge_images(an, p, ia.copy(), ib)
        assert_same(merged, whole, "merge " + an)
        for i in range(4):
            assert np.array_equal(O.merge_images(ALGO[an], p, ia[i], ib[i]), whole[i])
        acc = ctx.sketch_batch(an, k, p, 42, sb, ob, gb, flags=lash_amd.F_ACCUMULATE, out=ia.copy())
        assert_same(acc, whole, "accumulate " + an)
        again = ctx.sketch_batch(an, k, p, 42, sb, ob, gb, flags=lash_amd.F_ACCUMULATE, out=acc.copy())
        assert_same(again, whole, "idempotent " + an)


def test_device_entries_and_synth_generator(ctx):
    import torch
    import lash_amd
    n_g, L = 6, 100_003
    d_seq = torch.empty(n_g * L, dtype=torch.uint8, device="cuda:0")
    ctx.synth_genomes_device(1000, n_g, L, d_seq)
    ctx.synchronize()
    host = d_seq.cpu().numpy()
    for g in range(n_g):
        assert np.array_equal(host[g * L:(g + 1) * L], O.synth_genome(1000 + g, L)), g
    rec_off = (np.arange(n_g + 1, dtype=np.uint64) * L)
    goff = np.arange(n_g + 1, dtype=np.uint64)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to("cuda:0")
    ib = lash_amd.image_bytes("hmh")
    d_out = torch.zeros(n_g * ib, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()                 # torch's fill runs on its own stream
    ctx.enable_timing(True)
    ctx.sketch_batch_device("hmh", 16, 0, 42, d_seq, d_rec, n_g, goff, rec_off, d_out)
    t = ctx.timing()
    want = oracle_images(0, 16, 0, 42, host, rec_off, goff)
    assert_same(d_out.cpu().numpy().reshape(n_g, ib), want, "device entry")
    assert t["kmers"] == n_g * (L - 15) and t["bases_last"] == n_g * L and t["calls"] == 1
    assert t["sketch_ms"] > 0 and (t["pack_ms"] > 0 or R.sole_on())     # (six 100 kbp genomes: the persistent kernel takes them all, nothing is uploaded for slices)
    # two-stage form: pack once, sketch with several parameter sets
    pk = ctx.pack_device(d_seq, d_rec, n_g, goff, rec_off)
    for an, k, p in (("hll", 21, 14), ("ull", 16, 12), ("hmh", 24, 0)):
        ibx = lash_amd.image_bytes(an, p)
        d_o = torch.zeros(n_g * ibx, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ctx.sketch_packed_device(an, k, p, 42, pk, d_o)
        ctx.synchronize()
        assert_same(d_o.cpu().numpy().reshape(n_g, ibx), oracle_images(ALGO[an], k, p, 42, host, rec_off, goff), an)
    pk.free()
    ctx.enable_timing(False)


def test_runs_on_callers_stream(ctx):
    import torch
    import lash_amd
    s = torch.cuda.Stream()
    ctx.set_stream(s)
    gs = [[O.synth_genome(3, 50000).tobytes()]]
    seq, off, goff = lash_amd.records_to_arrays(gs)
    got = ctx.sketch_batch("ull", 16, 12, 42, seq, off, goff)
    ctx.set_stream(None)
    assert_same(got, oracle_images(2, 16, 12, 42, seq, off, goff), "stream")


def test_kmer_census_with_messy_records(ctx):
    import lash_amd
    rng = random.Random(3)
    gs = messy_genomes(rng, 9)
    seq, off, goff = lash_amd.records_to_arrays(gs)
    for k in (1, 7, 16, 21, 32):
        want = sum(len(O.record_kmers(r, k)) for g in gs for r in g)
        ctx.enable_timing(True)
        ctx.sketch_batch("hmh", k, 0, 42, seq, off, goff)
        assert ctx.timing()["kmers"] == want, k
    ctx.enable_timing(False)


def test_partial_sketches_of_one_input_merge_across_ranks():
    """configs[4] shape on the GPU path: two halves of a read set sketched separately (as two ranks would), all-gathered
    over RCCL (world_size 1 here: one GPU per box) and folded with lash_merge_images_device == the sketch of the whole."""
    import socket
    import torch
    import torch.distributed as dist
    import lash_amd
    from lash_amd.shard import merge_partial_images
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        c = lash_amd.Context(0, stream=torch.cuda.current_stream())
        g = O.synth_genome(55, 600_000).tobytes()
        reads = [g[i:i + 150] for i in range(0, len(g) - 150, 97)]
        half = len(reads) // 2
        for an, k, p in (("hmh", 16, 0), ("hll", 21, 12), ("ull", 16, 11)):
            parts = []
            for chunk in (reads[:half], reads[half:]):
                seq, off, goff = lash_amd.records_to_arrays([chunk])
                parts.append(torch.from_numpy(c.sketch_batch(an, k, p, 42, seq, off, goff)).cuda())
            ib = parts[0].shape[1]
            merged = merge_partial_images(parts[0], lambda d, s_: c.merge_images_device(an, p, d, s_, d.shape[0]))
            c.merge_images_device(an, p, merged, parts[1], 1)           # the "other rank's" partial
            torch.cuda.synchronize()
            seq, off, goff = lash_amd.records_to_arrays([reads])
            assert_same(merged.cpu().numpy(), oracle_images(ALGO[an], k, p, 42, seq, off, goff), "merged " + an)
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("ull", 16, 12), ("hll", 21, 16)])
def test_one_input_with_very_many_slices(ctx, an, k, p):
    """A metagenome-sized input (BASELINE configs[4] shape, scaled): one 'genome' of 150-bp reads cut into far more than 64
    work-item slices, whose partial sketches are folded in groups before the per-genome finalize (and, for hll p=16, in
    two bucket-space passes as well), next to a small genome in the same batch."""
    import lash_amd
    g = O.synth_genome(77, 40_000_000)
    reads = [g[i:i + 150].tobytes() for i in range(0, len(g) - 150, 150)]
    reads[1000] = b"ACGTN" * 30
    small = [O.synth_genome(78, 40_000).tobytes()]
    seq, off, goff = lash_amd.records_to_arrays([reads, small])
    want = oracle_images(ALGO[an], k, p, 42, seq, off, goff)
    ctx.enable_timing(True)
    got = ctx.sketch_batch(an, k, p, 42, seq, off, goff)
    tm = ctx.timing()
    ctx.enable_timing(False)
    assert tm["sketch_workgroups"] > 64 * (2 if p == 16 else 1)      # more than 64 slices per pass: the grouped fold runs
    assert tm["kmers"] == (len(reads) - 1) * (150 - k + 1) + sum(len(O.record_kmers(r, k)) for r in (reads[1000], small[0]))
    assert_same(got, want, an)
    assert_same(ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT), want, an + " pack-first")


def test_async_host_entry_overlaps_and_matches(ctx):
    """lash_sketch_batch_async: several batches in flight over the context's two staging slots, page-locked and pageable
    buffers, accumulate included; after lash_ctx_synchronize every image equals the oracle's."""
    import lash_amd
    rng = random.Random(99)
    batches = []
    for b in range(5):
        gs = messy_genomes(rng, 6, max_len=20_000)
        seq, off, goff = lash_amd.records_to_arrays(gs)
        batches.append((seq, off, goff))
    for an, k, p in (("hmh", 16, 0), ("ull", 21, 12)):
        ib = lash_amd.image_bytes(an, p)
        keep, outs = [], []
        for seq, off, goff in batches:
            ps, po = lash_amd.PinnedArray(max(len(seq), 1)), lash_amd.PinnedArray(len(off) * 8, np.uint64)
            ps.array[:len(seq)] = seq
            po.array[:] = off
            out = lash_amd.PinnedArray(6 * ib)
            keep.append((ps, po, out))
            ctx.sketch_batch_async(an, k, p, 42, ps.array[:len(seq)], po.array, goff, out.array)
            outs.append(out)
        ctx.synchronize()
        for (seq, off, goff), out in zip(batches, outs):
            assert_same(out.array.reshape(6, ib), oracle_images(ALGO[an], k, p, 42, seq, off, goff), "async " + an)
        # pageable memory + accumulate: batch 1's records unioned into batch 0's images
        acc = outs[0].array.reshape(6, ib).copy()
        s1, o1, g1 = batches[1]
        ctx.sketch_batch_async(an, k, p, 42, s1, o1, g1, acc.reshape(-1), flags=lash_amd.F_ACCUMULATE)
        ctx.synchronize()
        s0, o0, g0 = batches[0]
        want = O.merge_images(ALGO[an], p, oracle_images(ALGO[an], k, p, 42, s0, o0, g0)[0], oracle_images(ALGO[an], k, p, 42, s1, o1, g1)[0])
        assert np.array_equal(acc[0], want)


@pytest.mark.parametrize("flags", [0, 4])                  # direct route / F_NO_DIRECT
def test_runs_of_empty_and_tiny_records_share_break_words(ctx, flags):
    """Many record starts inside one 32-byte stretch: runs of empty records (any number of starts at ONE position) and
    1..3-base records.  The break bitmap of the direct route is written one word per "head" record looking 32 records
    ahead, with an atomic fallback beyond that (brk_bytes_kernel): both branches, at word boundaries too."""
    import lash_amd
    rng = random.Random(4242)
    acgt = lambda n: bytes(rng.choice(b"ACGT") for _ in range(n))
    gs = []
    for n_empty in (1, 30, 31, 32, 33, 64, 200):
        gs.append([acgt(100 + n_empty)] + [b""] * n_empty + [acgt(90)] + [b""] * (n_empty // 2) + [acgt(3), acgt(40)])
    gs.append([acgt(rng.choice([0, 1, 1, 2, 3, 20])) for _ in range(400)])                 # starts packed 1-3 bytes apart
    gs.append([acgt(31)] + [acgt(1)] * 70 + [acgt(64)] + [b""] * 40 + [acgt(1)] * 33 + [acgt(25)])
    gs.append([b""] * 50 + [acgt(60)] + [b""] * 50)                                        # empty records first and last
    seq, off, goff = lash_amd.records_to_arrays(gs)
    for an, k, p in (("hmh", 16, 0), ("ull", 5, 10), ("hll", 21, 12), ("hmh", 2, 0)):
        got = ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=flags)
        want = oracle_images(ALGO[an], k, p, 42, seq, off, goff)
        assert_same(got, want, "%s k=%d flags=%d" % (an, k, flags))


def test_a_genome_whose_bin_lists_outgrow_the_budget_takes_the_global_table(ctx, monkeypatch):
    """the pre-check of a binned launch estimated 5 bytes of list per input byte where bins_prepare sizes ~6.75 at
    ULL p = 22, so a genome in between passed the first, failed the second and the call returned LASH_ELIMIT instead of taking the
    table-in-global-memory plan.  With LASH_BINS_MB=1024 the window is 147 .. 185 MB at p = 22: a 160 Mbp genome now sketches (planned
    again without bins) and equals the oracle."""
    import torch
    import lash_amd
    monkeypatch.setenv("LASH_BINS_MB", "1024")
    L = 160_000_000
    d_seq = torch.empty(L, dtype=torch.uint8, device="cuda")
    ctx.synth_genomes_device(77, 1, L, d_seq)
    rec_off = np.array([0, L], dtype=np.uint64)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
    goff = np.array([0, 1], dtype=np.uint64)
    ib = lash_amd.image_bytes("ull", 22)
    d_img = torch.zeros(ib, dtype=torch.uint8, device="cuda")
    ctx.sketch_batch_device("ull", 16, 22, 42, d_seq, d_rec, 1, goff, rec_off, d_img)        # raised LASH_ELIMIT before
    ctx.synchronize()
    host = O.synth_genome(77, L)
    want = oracle_images(ALGO["ull"], 16, 22, 42, host, rec_off, goff)
    assert_same(d_img.cpu().numpy().reshape(1, ib), want, "ull p=22, 160 Mbp, LASH_BINS_MB=1024")
