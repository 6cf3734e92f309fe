"""GPU side of the layout switches (include/lash_gfx950.h `lash_layout`): for every alternative the
oracle implements, the HIP path with the same layout set on the context must give the oracle's bytes — through the direct
(ASCII-reading) route, the pack-first route, the raw-file route, the merge entry and the dist-side pair kernels.
This is what makes a mismatch against real `lash` images (tools/ref_probe) a field change instead of a code change."""
import random
import zlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ALGO = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}

SPECS = [
    "codes=ACTG",
    "codes=TGCA,kmer=lsb",
    "kmer=lsb",
    "hmh_x=low,hmh_reg=be,hmh_hdr=l",
    "hll_bucket=high,hll_hdr=pzsal",
    "hll_hdr=PZ,ull_hdr=pL,hmh_hdr=p",
    "codes=GATC,ull_hdr=,hmh_hdr=Q,hll_bucket=high,kmer=lsb",
]


@pytest.fixture(scope="module")
def ctx():
    import lash_amd
    c = lash_amd.Context(0)
    yield c
    c.set_layout(None)
    c.close()


def _genomes(rng, clean):
    gs = []
    for g in range(5):
        recs = []
        for _ in range(rng.randint(1, 4)):
            n = rng.randint(0, 6000)
            s = "".join(rng.choice("ACGT") for _ in range(n))
            if not clean and n > 100 and rng.random() < 0.7:
                cut = rng.randint(0, n - 50)
                s = s[:cut] + rng.choice(["N" * rng.randint(1, 40), "acgtn", "RY"]) + s[cut:]
            recs.append(s.encode())
        gs.append(recs)
    gs.append([O.synth_genome(77, 300_000).tobytes()])              # several slices
    return gs


@pytest.mark.parametrize("spec", SPECS)
@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hmh", 11, 0), ("hmh", 31, 0), ("hll", 21, 14), ("hll", 16, 16),
                                    ("ull", 16, 12), ("ull", 32, 16), ("ull", 9, 19)])
def test_layout_alternatives_match_oracle(ctx, spec, an, k, p):
    import lash_amd
    lay = O.parse_layout(spec)
    ctx.set_layout(spec)
    try:
        seed = zlib.crc32(repr((spec, an, k, p)).encode())
        rng = random.Random(seed)
        for clean in (True, False):
            seq, off, goff = lash_amd.records_to_arrays(_genomes(rng, clean))
            want = O.sketch_genomes(ALGO[an], k, p, 42, seq, off, goff, threads=8, layout=lay)
            for flags in (0, lash_amd.F_NO_DIRECT):
                got = ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=flags)
                assert got.shape == want.shape and np.array_equal(got, want), (spec, an, k, p, clean, flags, seed)
        # union of serialized sketches through the same layout (header offsets, HMH byte order, HLL header fields)
        a, b = want[:3].copy(), want[3:6].copy()
        m = ctx.merge_images(an, p, a.copy(), b)
        for i in range(3):
            assert np.array_equal(m[i], O.merge_images(ALGO[an], p, a[i], b[i], layout=lay)), (spec, an, i)
    finally:
        ctx.set_layout(None)


@pytest.mark.parametrize("spec", ["codes=ACTG,hmh_hdr=l,hmh_reg=be", "hll_bucket=high,hll_hdr=pzsal,kmer=lsb"])
def test_layout_raw_files_and_pair_kernels(ctx, spec):
    lay = O.parse_layout(spec)
    ctx.set_layout(spec)
    try:
        rng = random.Random(zlib.crc32(spec.encode()))
        files = []
        for f in range(4):
            body = "".join(rng.choice("ACGT") for _ in range(20_000))
            lines = "\n".join(body[i:i + 70] for i in range(0, len(body), 70))
            files.append((">f%d\n%s\n>second\nACGTNNNNACGTACGTACGTTTGACCA\n" % (f, lines)).encode())
        for an, k, p in (("hmh", 16, 0), ("hll", 21, 12), ("ull", 16, 10)):
            got = ctx.sketch_files_raw(an, k, p, 42, files)
            want = O.sketch_files(ALGO[an], k, p, 42, files, threads=4, layout=lay)
            assert np.array_equal(got, want), (spec, an)
        # dist side: pair statistics read registers behind the layout's header; HMH counts do not depend on byte order
        hm = O.sketch_files(O.HMH, 16, 0, 42, files, layout=lay)
        c, n = ctx.hmh_pair_counts(hm, hm)
        hb = O.header_bytes(O.HMH, lay)
        ra = np.ascontiguousarray(hm[:, hb:]).view(">u2" if lay.hmh_reg_be else "<u2").astype(np.int64)
        for i in range(4):
            for j in range(4):
                assert c[i, j] == int(((ra[i] != 0) & (ra[i] == ra[j])).sum()) and n[i, j] == int(((ra[i] != 0) | (ra[j] != 0)).sum())
        assert hm.shape[1] == hb + 32768
        hl = O.sketch_files(O.HLL, 21, 12, 42, files, layout=lay)
        zero, usum = ctx.hll_pair_union_stats(12, hl, hl)
        hh = O.header_bytes(O.HLL, lay)
        regs = hl[:, hh:].astype(np.int64)
        for i in range(4):
            for j in range(4):
                u = np.maximum(regs[i], regs[j])
                assert zero[i, j] == int((u == 0).sum()) and abs(usum[i, j] - float(np.sum(2.0 ** -u.astype(np.float64)))) < 1e-9
    finally:
        ctx.set_layout(None)


# Round 6: every rule alternative is a compile-time variant of EVERY kernel family, not a packed-only slow route.
# One alternative at a time and all of them together, through each route the library can take for a genome: the persistent small-genome
# kernel (the default for genomes <= LASH_SOLE_MAX), the sliced direct kernel (with and without deferred signatures), the compacting
# stream kernel, the pack-first route, large register tables as bytes and as bins.
RULE_SPECS = ["hmh_x=low", "kmer=lsb", "hll_bucket=high", "codes=GATC,kmer=lsb,hll_bucket=high,hmh_x=low"]


def _route_genomes(rng):
    gs = _genomes(rng, clean=False)
    gs.append([O.synth_genome(78, 1_300_000).tobytes()])             # work items long enough for the deferring launch's default threshold
    soft = bytearray(O.synth_genome(79, 400_000).tobytes())           # a soft-masked stretch and an N gap: junction walks, dense tiles, stream
    soft[100_000:160_000] = bytes(soft[100_000:160_000]).lower()
    soft[300_000:300_700] = b"N" * 700
    gs.append([bytes(soft[:250_000]), bytes(soft[250_000:])])
    return gs


@pytest.mark.sole
@pytest.mark.parametrize("spec", RULE_SPECS)
@pytest.mark.parametrize("an,k,p", [("hmh", 16, 0), ("hmh", 9, 0), ("hmh", 27, 0), ("hll", 21, 14), ("hll", 16, 10), ("hll", 12, 16),
                                    ("ull", 16, 12), ("ull", 31, 15)])
def test_rule_alternatives_through_every_route(ctx, spec, an, k, p, monkeypatch):
    import lash_amd
    lay = O.parse_layout(spec)
    ctx.set_layout(spec)
    try:
        seed = zlib.crc32(repr(("routes", spec, an, k, p)).encode())
        seq, off, goff = lash_amd.records_to_arrays(_route_genomes(random.Random(seed)))
        want = O.sketch_genomes(ALGO[an], k, p, 42, seq, off, goff, threads=8, layout=lay)
        routes = [("default (small genomes through the persistent kernel)", 0, {}),
                  ("sliced direct kernel", lash_amd.F_NO_SOLE, {}),
                  ("sliced direct kernel, every HyperMinHash launch deferring", lash_amd.F_NO_SOLE, {"LASH_DEFER_MIN": "0"}),
                  ("stream kernel", lash_amd.F_STREAM_ONLY, {}),
                  ("stream kernel, deferring", lash_amd.F_STREAM_ONLY, {"LASH_DEFER_MIN": "0"}),
                  ("pack first", lash_amd.F_NO_DIRECT | lash_amd.F_NO_SOLE, {}),
                  ("pack first, persistent kernel on packed words", lash_amd.F_NO_DIRECT, {})]
        if p >= 15:
            routes.append(("bins instead of byte tables", lash_amd.F_NO_SOLE, {"LASH_NO_BYTES": "1"}))
        for name, flags, envs in routes:
            if "LASH_DEFER_MIN" in envs and an != "hmh":
                continue
            with monkeypatch.context() as m:
                for key, v in envs.items():
                    m.setenv(key, v)
                ctx.enable_timing(True)
                got = ctx.sketch_batch(an, k, p, 42, seq, off, goff, flags=flags)
                tm = ctx.timing()
                ctx.enable_timing(False)
            assert got.shape == want.shape and np.array_equal(got, want), (spec, an, k, p, name, seed)
            if envs.get("LASH_DEFER_MIN") == "0" and not (flags & lash_amd.F_STREAM_ONLY):
                assert tm["defer_launches"] >= 1, (spec, name, tm)      # x = low defers like the default (round 6)
            if flags == 0 and p <= 14:                                   # (a table the persistent kernel's LDS budget holds)
                assert tm["sole_launches"] >= 1, (spec, name, tm)       # no layout keeps a genome from the persistent kernel any more
            if flags == lash_amd.F_NO_SOLE and not envs:
                assert tm["direct_launches"] >= 1, (spec, name, tm)     # ... or from the ASCII-reading kernels
    finally:
        ctx.set_layout(None)


@pytest.mark.parametrize("spec,an,k,p", [("hmh_x=low", "hmh", 16, 0), ("kmer=lsb", "hmh", 16, 0), ("kmer=lsb", "ull", 16, 12),
                                         ("hll_bucket=high", "hll", 21, 14), ("kmer=lsb", "hll", 21, 14)])
def test_rule_alternatives_full_size_spot_check(ctx, spec, an, k, p):
    """40 (hmh: 300) x 5 Mbp per alternative, ASCII resident in HBM -> the judged kernels' variants at the size bench.py runs them
    (sliced direct kernel, HyperMinHash deferring by itself); three genomes against the oracle, the census against L - k + 1."""
    import torch
    import lash_amd
    G, L = (300 if an == "hmh" else 40), 5_000_000              # (hmh: enough genomes for work items of >= 0.6 Mbp, the deferring launch's threshold)
    lay = O.parse_layout(spec)
    ctx.set_layout(spec)
    try:
        d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda:0")
        ctx.synth_genomes_device(3000, G, L, d_seq)
        rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
        goff = np.arange(G + 1, dtype=np.uint64)
        d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
        ib = ctx.image_bytes(an, p)
        d_img = torch.zeros(G * ib, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ctx.enable_timing(True)
        ctx.sketch_batch_device(an, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
        tm = ctx.timing()
        ctx.enable_timing(False)
        assert tm["kmers"] == G * (L - k + 1) and tm["direct_launches"] == 1, tm
        if an == "hmh" and "LASH_DEFER_MIN" not in __import__("os").environ:
            assert tm["defer_launches"] == 1, tm
        img = d_img.view(G, ib).cpu().numpy()
        for g in (0, 17, G - 1):
            want = O.sketch_genomes(ALGO[an], k, p, 42, O.synth_genome(3000 + g, L), np.array([0, L], np.uint64), np.array([0, 1], np.uint64),
                                    layout=lay)[0]
            assert np.array_equal(img[g], want), (spec, an, g)
    finally:
        ctx.set_layout(None)


def test_a_packed_batch_belongs_to_the_code_table_it_was_packed_under(ctx):
    """Round 6: kmer_lsb_first is the complemented code table, so packed 2-bit words depend on it (as they always did on `codes=`).  Sketching a batch
    packed under another table would silently give another sketch: LASH_EINVAL instead.  Header templates and the hash half may change freely."""
    import torch
    import lash_amd
    seq, off, goff = lash_amd.records_to_arrays([[O.synth_genome(5, 50_000).tobytes()], [O.synth_genome(6, 30_000).tobytes()]])
    d_seq = torch.from_numpy(seq).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    gbo = off[goff.astype(np.int64)]
    try:
        ctx.set_layout(None)
        pk = ctx.pack_device(d_seq, d_off, len(off) - 1, goff, gbo)
        d_img = torch.zeros(2 * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device="cuda")
        ctx.sketch_packed_device("hmh", 16, 0, 42, pk, d_img)
        ctx.set_layout("hmh_x=low,hmh_hdr=l")                              # same k-mers: allowed
        d_img2 = torch.zeros(2 * ctx.image_bytes("hmh"), dtype=torch.uint8, device="cuda")
        ctx.sketch_packed_device("hmh", 16, 0, 42, pk, d_img2)
        ctx.synchronize()
        want = O.sketch_genomes(O.HMH, 16, 0, 42, seq, off, goff, layout=O.parse_layout("hmh_x=low,hmh_hdr=l"))
        assert np.array_equal(d_img2.cpu().numpy().reshape(2, -1), want)
        for spec in ("kmer=lsb", "codes=ACTG"):
            ctx.set_layout(spec)
            with pytest.raises(lash_amd.LashError) as e:
                ctx.sketch_packed_device("hmh", 16, 0, 42, pk, d_img)
            assert e.value.code == lash_amd.EINVAL
        pk.free()
    finally:
        ctx.set_layout(None)
