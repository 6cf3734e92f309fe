#!/usr/bin/env python3
"""tests/fuzz_gpu.py [iterations] [seed] — randomized GPU-vs-oracle parity runs (not collected by pytest; run it on the
GPU box when a kernel changed: `python tests/fuzz_gpu.py 300`).  Random sketch type, k, p, seed, flags and batch shapes:
clean / dirty / mixed genomes, many or few records, lengths around lane, tile and slice boundaries."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lash_amd
import oracle_lib as O
import fuzz_knobs

ALGO = {"hmh": 0, "hll": 1, "ull": 2}


def random_genome(rng):
    kind = rng.random()
    n_rec = rng.choice([1, 1, 1, 2, 3, rng.randint(4, 40)])
    recs = []
    for _ in range(n_rec):
        L = rng.choice([rng.randint(0, 120), rng.randint(0, 5000), rng.randint(0, 200_000),
                        rng.choice([63, 64, 65, 95, 96, 97, 2047, 2048, 2049, 4096, 16384, 16385, 32768]),
                        rng.choice([262_144, 262_145, rng.randint(300_000, 2_500_000)]) if rng.random() < 0.3 else 1000])   # several slices
        s = bytearray(O.synth_genome(rng.randint(0, 10**6), max(L, 1)).tobytes()[:L])
        if kind < 0.3:
            pass                                            # clean
        elif kind < 0.45 and L:
            # block dirt: soft-masked stretches and gaps of 50 bytes .. 200 kb (whole wave-tiles deleted, tiles that end inside a
            # run, runs that span slices and records) — what the in-kernel compaction (dense_tile) exists for
            pos = rng.randrange(L)
            while pos < L:
                run = min(rng.choice([50, 300, 1000, 4096, 5000, 12_000, 40_000, 200_000]), L - pos)
                fill = rng.choice([b"N", b"n", None])
                if fill is None:
                    s[pos:pos + run] = bytes(s[pos:pos + run]).lower()
                else:
                    s[pos:pos + run] = fill * run
                pos += run + rng.choice([1, 5, 15, 16, 31, 64, 700, 4096, 9000, 60_000])
        elif kind < 0.7 and L:
            for _ in range(rng.randint(1, 3)):              # sparse dirt
                i = rng.randrange(L)
                s[i:i + rng.choice([1, 1, 2, 10, 100])] = rng.choice([b"N", b"n", b"a", b"-", b"R"]) * min(rng.choice([1, 1, 2, 10, 100]), L - i)
        elif L:
            for i in range(L):                              # dense dirt
                if rng.random() < 0.2:
                    s[i] = rng.choice(b"NnacgtRYKM-*")
        recs.append(bytes(s))
    return recs


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = lash_amd.Context(0)
    only = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None     # one iteration again, with the differing bytes printed
    for it in range(iters):
        if only is not None and it != only:
            continue
        rng = random.Random(seed0 * 100003 + it)
        knobs = fuzz_knobs.set_sole(random.Random(seed0 * 1000003 + it))    # which genomes go to the persistent small-genome kernel (FUZZ_SOLE)
        an = rng.choice(["hmh", "hll", "ull"])
        if os.environ.get("FUZZ_ALGO"):                     # e.g. FUZZ_ALGO=hmh LASH_DEFER_MIN=0: every direct launch defers its signatures
            an = os.environ["FUZZ_ALGO"]
        k = rng.choice([rng.randint(1, 32), 16, 21, 31, 32])
        p = 0 if an == "hmh" else rng.choice([rng.randint(4, 16)] if an == "hll" else [rng.randint(3, 20), rng.randint(3, 14)])
        if os.environ.get("FUZZ_P") and an != "hmh":      # e.g. FUZZ_P=15,16,17: the byte tables (hll: p <= 16)
            p = min(rng.choice([int(x) for x in os.environ["FUZZ_P"].split(",")]), 16 if an == "hll" else 26)
        seed = rng.choice([0, 42, rng.getrandbits(64)])
        # round 6: the unpinned register / k-mer rules are compile-time variants of every kernel family — drawn per iteration (FUZZ_LAYOUT pins one;
        # FUZZ_LAYOUT=default pins the default)
        lrng = random.Random(seed0 * 7000003 + it)
        spec = lrng.choice([None, None, None, "hmh_x=low", "kmer=lsb", "hll_bucket=high", "codes=GATC,kmer=lsb,hll_bucket=high,hmh_x=low"])
        if os.environ.get("FUZZ_LAYOUT"):
            spec = None if os.environ["FUZZ_LAYOUT"] == "default" else os.environ["FUZZ_LAYOUT"]
        ctx.set_layout(spec)
        lay = O.parse_layout(spec) if spec else None
        knobs += " layout=%s" % (spec or "default")
        flags = rng.choice([0, 0, lash_amd.F_NO_DIRECT, lash_amd.F_STREAM_ONLY]) | (lash_amd.F_HMH_X_LOW if an == "hmh" and rng.random() < 0.2 else 0)
        gs = [random_genome(rng) for _ in range(rng.randint(1, 12))]
        seq, off, goff = lash_amd.records_to_arrays(gs)
        mode = rng.choice(["host", "host", "accumulate", "device", "packed"])
        ctx.enable_timing(True)
        if mode == "host" or len(seq) == 0:
            got = ctx.sketch_batch(an, k, p, seed, seq, off, goff, flags=flags)
        elif mode == "accumulate":
            # every genome's records in two calls: the first ones, then the rest unioned into the same images
            cut = [rng.randint(0, len(g)) for g in gs]
            s1, o1, g1 = lash_amd.records_to_arrays([g[:c] for g, c in zip(gs, cut)])
            s2, o2, g2 = lash_amd.records_to_arrays([g[c:] for g, c in zip(gs, cut)])
            got = ctx.sketch_batch(an, k, p, seed, s1, o1, g1, flags=flags)
            got = ctx.sketch_batch(an, k, p, seed, s2, o2, g2, flags=flags | lash_amd.F_ACCUMULATE, out=got)
        else:
            import torch
            d_seq = torch.from_numpy(seq).cuda()
            d_off = torch.from_numpy(off.astype(np.int64)).cuda()
            gbo = off[goff.astype(np.int64)]
            d_img = torch.zeros(len(gs) * ctx.image_bytes(an, p), dtype=torch.uint8, device="cuda")
            if mode == "device":
                ctx.sketch_batch_device(an, k, p, seed, d_seq, d_off, len(off) - 1, goff, gbo, d_img, flags=flags)
            else:
                pk = ctx.pack_device(d_seq, d_off, len(off) - 1, goff, gbo)
                ctx.sketch_packed_device(an, k, p, seed, pk, d_img, flags=flags & ~(lash_amd.F_NO_DIRECT | lash_amd.F_STREAM_ONLY))
                ctx.sketch_packed_device(an, k, p, seed, pk, d_img, flags=(flags & ~(lash_amd.F_NO_DIRECT | lash_amd.F_STREAM_ONLY)) | lash_amd.F_ACCUMULATE)   # idempotent
            ctx.synchronize()
            # the device entries only FLAG a HyperLogLog genome whose `sum` the incremental rule rounds (a register above 53 - p: one
            # k-mer in 2^37 at p = 16 — it=187 of seed 44 met one, xxh3_64 = 0x0000000002025fbf); the caller asks for the replay
            census_before_replay = ctx.timing()["kmers"]
            if an == "hll" and ctx.hll_inexact_sums():
                ctx.hll_replay_sums_device(k, p, seed, d_seq, d_off, len(off) - 1, goff, d_img, flags=flags & ~(lash_amd.F_NO_DIRECT | lash_amd.F_STREAM_ONLY))
            got = d_img.cpu().numpy().reshape(len(gs), -1)
            if mode == "packed":
                pk.free()
        kmers = ctx.timing()["kmers"] if mode in ("host", "accumulate") or len(seq) == 0 else census_before_replay
        ctx.enable_timing(False)
        if mode == "packed":
            kmers //= 2                                      # sketched twice
        want = O.sketch_genomes(ALGO[an], k, p, seed, seq, off, goff, threads=8, hmh_x_is_low=1 if flags & lash_amd.F_HMH_X_LOW else 0, layout=lay)
        want_kmers = sum(len(O.record_kmers(r, k)) for g in gs for r in g)
        if not np.array_equal(got, want) or kmers != want_kmers:
            bad = sorted({int(r) for r in np.argwhere(got != want)[:, 0]}) if got.shape == want.shape else "shape"
            print("MISMATCH [" + knobs + "] it=%d mode=%s %s k=%d p=%d seed=%d flags=%d genomes=%s census %d vs %d" % (it, mode, an, k, p, seed, flags, bad, kmers, want_kmers))
            if only is not None and got.shape == want.shape:
                d = np.argwhere(got != want)
                print("  %d bytes differ; first: %s" % (len(d), [(int(g), int(o), int(got[g, o]), int(want[g, o])) for g, o in d[:24]]))
                print("  genome sizes:", [sum(len(r) for r in g) for g in gs], "records:", [len(g) for g in gs])
                if an == "hll":
                    for g in sorted({int(g) for g, _ in d}):
                        hdr = lash_amd.image_bytes(an, p) - (1 << p)
                        regs = want[g, hdr:].astype(np.int64)
                        print("  genome %d: sum got %r want %r (hex %s / %s); registers max %d, histogram of the top: %s; exact sum %r" % (
                            g, got[g, 16:24].view(np.float64)[0], want[g, 16:24].view(np.float64)[0], got[g, 16:24].tobytes().hex(), want[g, 16:24].tobytes().hex(),
                            regs.max(), np.bincount(regs)[-6:].tolist(), float(sum(np.ldexp(1.0, -int(r)) * int(c) for r, c in enumerate(np.bincount(regs))))))
            sys.exit(1)
    print("fuzz ok: %d iterations from seed %d" % (iters, seed0))


if __name__ == "__main__":
    main()
