"""The one known corner where the HIP path's HyperLogLog image can differ from the reference's by construction: the `sum`
field when a register exceeds 53 - p (include/lash_gfx950.h: lash_ctx_hll_inexact_sums).

streaming_algorithms updates `sum` per k-mer in f64; up to 53 - p every update is exact, so the histogram sum the kernels
write IS the reference's value (all other tests).  Above it the incremental value depends on rounding order.  A k-mer that
gets there turns up once per 2^(52-p) hashes; tools/find_hll_corner.py found the ones below with the sketch kernel itself
(seed 42, k = 21; profiles/r02/hll_corner_kmers.txt).  What is asserted here:
  * the oracle (which restates the incremental rule) agrees these k-mers have rank 38..40;
  * registers, `zero` and every byte outside `sum` are identical between the HIP path and the oracle, on every route;
  * the kernels' `sum` is the correctly rounded exact sum, the genome is reported, clean genomes are not;
  * the divergence is real (a case where the oracle's incremental value differs is pinned) and bounded by the sub-grid
    terms themselves: < 2^(p-52) absolute, ~1e-14 relative;
  * round 4: the REPLAY (lash_hll_replay_sums_device; built into lash_sketch_batch and lash_sketch_files_raw) turns the 8 bytes into
    the oracle's — whole images byte-identical in the corner too; only accumulating calls stay flagged."""
import struct
from fractions import Fraction

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["sliced kernels", "persistent kernel", "persistent kernel, one workgroup"])
def _route(request, monkeypatch, _sole_mode):
    """Every test on both kernel families: the sliced kernels (tests/conftest.py pins LASH_SOLE_MAX=0 for tests not marked `sole`) and — the
    test genomes are small: what the library does with them by default — the persistent small-genome kernel, whose flush sets the corner flag
    itself and whose launches the replay's prefix probes are (p = 14: its table is the largest that kernel takes; p = 16 stays sliced).
    (Runs after conftest's `_sole_mode`, which it names as an argument.)"""
    import os
    if request.param != "sliced kernels":
        monkeypatch.delenv("LASH_SOLE_MAX", raising=False)
        if request.param.endswith("one workgroup"):
            monkeypatch.setenv("LASH_SOLE_WGS", "1")
    yield


CORNER = {  # k-mer -> rank (= leading zeros of the hash above the bucket bits + 1), the same for p = 14 and p = 16
    "CTGAGTGTGTCAGGCGTCATT": 40,
    "CGCTCAGTTGGAACGGTGCTT": 39,
    "CACCATGCTATGTGCATGACC": 40,
    "TTACGAATCATAAGGTCGATA": 38,
}


def _oracle(p, g):
    return O.sketch_genomes(O.HLL, 21, p, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0]


def _exact_sum(regs):
    s = sum(Fraction(1, 1 << int(r)) for r in regs)
    return float(s)                                  # Fraction -> float is correctly rounded


def test_corner_kmers_reach_the_stated_rank_in_the_oracle():
    for km, rho in CORNER.items():
        for p in (14, 16):
            img = _oracle(p, np.frombuffer(km.encode(), np.uint8))
            assert int(img[33:].max()) == rho and int((img[33:] != 0).sum()) == 1


@pytest.mark.parametrize("p", [14, 16])
@pytest.mark.parametrize("where", ["first", "middle"])
@pytest.mark.parametrize("size", [30_000, 3_000_000])          # one work item writes the image / slices + finalize
def test_corner_genomes_are_detected_and_their_sum_is_replayed(p, where, size):
    """Round 4: byte equality with the oracle IN the corner.  The device entry alone writes the correctly rounded
    exact sum and reports the genomes (as before); lash_hll_replay_sums_device then gives those genomes the reference's incremental
    value; the host entry lash_sketch_batch does both before it returns."""
    import torch
    import lash_amd
    kms = [km for km, rho in CORNER.items() if rho > 53 - p]
    assert kms
    genomes = []
    for i, km in enumerate(kms):
        rest = O.synth_genome(900 + i, size)
        kb = np.frombuffer(km.encode(), np.uint8)
        genomes.append(np.concatenate([kb, rest]) if where == "first" else np.concatenate([rest[:size // 2], kb, rest[size // 2:]]))
    genomes.append(O.synth_genome(950, size))                                  # a clean one between them
    genomes.insert(1, O.synth_genome(951, size // 3))
    corner_idx = [i for i, g in enumerate(genomes) if any(km.encode() in g.tobytes() for km in kms)]
    recs = [[g.tobytes()] for g in genomes]
    seq, off, goff = lash_amd.records_to_arrays(recs)
    want = np.stack([_oracle(p, g) for g in genomes])
    ib = want.shape[1]
    with lash_amd.Context(0) as ctx:
        d_seq, d_off = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        gbo = off[goff.astype(np.int64)]
        for flags in (0, lash_amd.F_NO_DIRECT):
            d_img = torch.zeros(len(genomes) * ib, dtype=torch.uint8, device="cuda")
            ctx.sketch_batch_device("hll", 21, p, 42, d_seq, d_off, len(off) - 1, goff, gbo, d_img, flags=flags)
            assert ctx.hll_inexact_sums() == corner_idx
            got = d_img.cpu().numpy().reshape(len(genomes), ib)
            assert np.array_equal(got[:, :16], want[:, :16]) and np.array_equal(got[:, 24:], want[:, 24:])   # all but `sum`
            for i in range(len(genomes)):
                gs = struct.unpack("<d", got[i, 16:24].tobytes())[0]
                ws = struct.unpack("<d", want[i, 16:24].tobytes())[0]
                assert gs == _exact_sum(got[i, 33:])                                           # correctly rounded exact sum
                if i not in corner_idx:
                    assert gs == ws                                                             # outside the corner: bit-identical
                else:
                    assert abs(gs - ws) <= 2.0 ** -(52 - p)              # at most the few terms that sit below the 2^(p-53) grid
            ctx.hll_replay_sums_device(21, p, 42, d_seq, d_off, len(off) - 1, goff, d_img, flags=flags)
            assert ctx.hll_inexact_sums() == []
            assert np.array_equal(d_img.cpu().numpy().reshape(len(genomes), ib), want), "replayed sums != the oracle's incremental sums"
            # the host entry does it by itself
            got = ctx.sketch_batch("hll", 21, p, 42, seq, off, goff, flags=flags)
            assert ctx.hll_inexact_sums() == [] and np.array_equal(got, want)
        # an accumulating call cannot be replayed (the registers already in the image are not this call's): still reported
        half = [[g.tobytes()[:len(g) // 2], g.tobytes()[len(g) // 2:]] for g in genomes]
        s1, o1, g1 = lash_amd.records_to_arrays([[h[0]] for h in half])
        s2, o2, g2 = lash_amd.records_to_arrays([[h[1]] for h in half])
        acc = ctx.sketch_batch("hll", 21, p, 42, s1, o1, g1)
        acc = ctx.sketch_batch("hll", 21, p, 42, s2, o2, g2, flags=lash_amd.F_ACCUMULATE, out=acc)
        assert set(ctx.hll_inexact_sums()) <= set(corner_idx)
        # a call without such genomes reports nothing
        s2, o2, g2 = lash_amd.records_to_arrays([[genomes[-1].tobytes()]])
        ctx.sketch_batch("hll", 21, p, 42, s2, o2, g2)
        assert ctx.hll_inexact_sums() == []
        ctx.sketch_batch("ull", 21, 12, 42, s2, o2, g2)
        assert ctx.hll_inexact_sums() == []


def test_replay_with_records_deleted_bytes_and_two_corner_kmers():
    """The replay cuts prefixes by BYTES: records, deleted bytes (N, lower case) and several corner k-mers in one genome — two in
    different buckets, and the same one twice — need nothing special."""
    import lash_amd
    p = 14
    kms = [np.frombuffer(km.encode(), np.uint8) for km, rho in CORNER.items() if rho > 53 - p]
    a, b, c = O.synth_genome(960, 400_000), O.synth_genome(961, 250_000), O.synth_genome(962, 120_000)
    b = b.copy(); b[1000:1300] = ord("N"); b[50_000:50_400] |= 0x20
    recs = [[a[:100_000].tobytes() + kms[0].tobytes() + a[100_000:].tobytes(), b"ACGT", b.tobytes()[:200_000] + kms[1].tobytes() + b"NN" + b.tobytes()[200_000:],
             c.tobytes() + kms[0].tobytes()],
            [O.synth_genome(963, 50_000).tobytes()]]
    seq, off, goff = lash_amd.records_to_arrays(recs)
    want = O.sketch_genomes(O.HLL, 21, p, 42, seq, off, goff)
    with lash_amd.Context(0) as ctx:
        got = ctx.sketch_batch("hll", 21, p, 42, seq, off, goff)
        assert ctx.hll_inexact_sums() == []
    assert np.array_equal(got, want)


def test_the_divergence_is_real_and_the_replay_closes_it():
    """p = 14, the rank-40 k-mer first: the reference's first update is 16383 + 2^-40, a tie that rounds to even and drops the
    term; 300 kbp later the sum is ~650 and 2^-40 is representable, so the exact sum (what the kernels write) has it — and the
    replayed value does not, like the reference's."""
    import torch
    import lash_amd
    g = np.concatenate([np.frombuffer(b"CTGAGTGTGTCAGGCGTCATT", np.uint8), O.synth_genome(5, 300_000)])
    want = _oracle(14, g)
    seq, off, goff = lash_amd.records_to_arrays([[g.tobytes()]])
    with lash_amd.Context(0) as ctx:
        d_seq, d_off = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
        d_img = torch.zeros(len(want), dtype=torch.uint8, device="cuda")
        ctx.sketch_batch_device("hll", 21, 14, 42, d_seq, d_off, 1, goff, off[goff.astype(np.int64)], d_img)
        assert ctx.hll_inexact_sums() == [0]
        got = d_img.cpu().numpy()
        gs = struct.unpack("<d", got[16:24].tobytes())[0]
        ws = struct.unpack("<d", want[16:24].tobytes())[0]
        assert gs == _exact_sum(got[33:]) and gs - ws == 2.0 ** -40
        assert np.array_equal(got[24:], want[24:]) and np.array_equal(got[:16], want[:16])
        ctx.hll_replay_sums_device(21, 14, 42, d_seq, d_off, 1, goff, d_img)
        assert np.array_equal(d_img.cpu().numpy(), want)


def test_cli_replays_whole_files_and_notes_streamed_ones(tmp_path):
    import os
    import subprocess
    import host_lib as H
    g = np.concatenate([np.frombuffer(b"CTGAGTGTGTCAGGCGTCATT", np.uint8), O.synth_genome(5, 3_000_000)])
    (tmp_path / "c.fa").write_bytes(b">c\n" + b"\n".join(g.tobytes()[i:i + 80] for i in range(0, len(g), 80)) + b"\n")
    (tmp_path / "d.fa").write_bytes(b">d\n" + O.synth_genome(6, 100_000).tobytes() + b"\n")
    (tmp_path / "l.txt").write_text("d.fa\nc.fa\n")
    for extra in ([], ["--stream-mb", "1"]):
        r = subprocess.run([H.CLI, "sketch", "-f", "l.txt", "-o", "h", "-a", "hll", "-p", "14", "-k", "21"] + extra, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        notes = [ln for ln in r.stderr.splitlines() if ln.startswith("note:")]
        imgs = np.frombuffer(H.zstd_read(str(tmp_path / "h_sketches.bin")), np.uint8).reshape(2, -1)
        assert np.array_equal(imgs[0], _oracle(14, O.synth_genome(6, 100_000)))
        # files sketched whole: the corner file's sum is replayed (round 4).  c.fa streamed in 1 MiB chunks with on-device accumulation
        # (round 5, lash_hll_replay_streamed_chunk): the chunk that lifts the register above 53 - p is replayed on top of the image
        # before it, the later chunks' exact net changes are carried — byte equality with the oracle's incremental sum, nothing to note
        assert notes == [] and np.array_equal(imgs[1], _oracle(14, g)), r.stderr


def test_streamed_file_with_corner_kmers_in_later_chunks(tmp_path):
    """Round 5: `lash sketch --stream-mb 1` on one file whose corner k-mers sit in DIFFERENT chunks — the 2nd and the
    4th of five — with clean chunks before, between and after: the chunk that lifts a register above 53 - p is replayed on top of the image
    before it (prefix sketches united with the earlier chunks' registers), every later chunk's exact net change is carried, a second
    corner k-mer two chunks on is replayed on top of the carried value.  The header equals the oracle's incremental sum byte for byte."""
    import subprocess
    import host_lib as H
    parts = [O.synth_genome(21, 1_300_000).tobytes(), b"CACCATGCTATGTGCATGACC", O.synth_genome(22, 2_000_000).tobytes(), b"CTGAGTGTGTCAGGCGTCATT",
             O.synth_genome(23, 1_200_000).tobytes()]
    g = np.frombuffer(b"".join(parts), np.uint8)
    (tmp_path / "s.fa").write_bytes(b">s\n" + b"\n".join(g.tobytes()[i:i + 70] for i in range(0, len(g), 70)) + b"\n")
    (tmp_path / "l.txt").write_text("s.fa\n")
    want = _oracle(14, g)
    for extra in (["--stream-mb", "1"], []):
        r = subprocess.run([H.CLI, "sketch", "-f", "l.txt", "-o", "h", "-a", "hll", "-p", "14", "-k", "21"] + extra, cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert [ln for ln in r.stderr.splitlines() if ln.startswith("note:")] == [], r.stderr
        img = np.frombuffer(H.zstd_read(str(tmp_path / "h_sketches.bin")), np.uint8)
        assert np.array_equal(img[24:], want[24:]) and np.array_equal(img[:16], want[:16])
        assert np.array_equal(img, want), (extra, struct.unpack("<d", img[16:24].tobytes())[0], struct.unpack("<d", want[16:24].tobytes())[0])
