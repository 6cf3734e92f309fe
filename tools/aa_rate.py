#!/usr/bin/env python3
"""tools/aa_rate.py — rate of the amino-acid sketch path (utils.rs:511-563) on a protein-database shape:
N proteins (default 10^6) of 50..2000 residues, uniform over the 20 letters, ONE sketch per G proteins (default 1000: a proteome),
resident in HBM; hmh k=7 and ull p=12 k=10.  Prints k-mers/s and residues/s per setting; the oracle checks one proteome.
GPU box.  N=... G=... python tools/aa_rate.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lash_amd

N, G = int(os.environ.get("N", 1_000_000)), int(os.environ.get("G", 1000))
rng = np.random.default_rng(7)
lens = rng.integers(50, 2001, size=N).astype(np.uint64)
rec_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
total = int(rec_off[-1])
letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
seq = letters[rng.integers(0, 20, size=total)]
n_gen = (N + G - 1) // G
goff = np.minimum(np.arange(n_gen + 1, dtype=np.uint64) * np.uint64(G), np.uint64(N))
gbo = rec_off[goff.astype(np.int64)]
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.from_numpy(seq).cuda()
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
print("%d proteins, %d residues (%.0f per protein), %d sketches of %d proteins" % (N, total, total / N, n_gen, G))
for algo, k, p in (("hmh", 7, 0), ("ull", 10, 12), ("hll", 7, 12)):
    ib = lash_amd.image_bytes(algo, p)
    d_img = torch.zeros(n_gen * ib, dtype=torch.uint8, device="cuda")
    kmers = int(np.maximum(lens.astype(np.int64) - k + 1, 0).sum())
    for _ in range(2):
        ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, N, goff, gbo, d_img, flags=lash_amd.F_AMINO)
    torch.cuda.synchronize()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    steps = int(os.environ.get("STEPS", 5))
    for _ in range(steps):
        ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, N, goff, gbo, d_img, flags=lash_amd.F_AMINO)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = ctx.timing()
    ctx.enable_timing(False)
    assert tm["kmers"] == kmers * steps, (tm["kmers"], kmers * steps)
    ok = ""
    if os.environ.get("CHECK", "1") == "1":
        import oracle_lib as O
        a, b = int(goff[0]), int(goff[1])
        s0, o0 = seq[int(rec_off[a]):int(rec_off[b])], (rec_off[a:b + 1] - rec_off[a]).astype(np.uint64)
        want = O.sketch_genomes(lash_amd.ALGOS[algo], k, p, 42, s0, o0, np.array([0, b - a], np.uint64), threads=4, amino=True)[0]
        ok = "  first sketch == oracle: %s" % bool(np.array_equal(d_img[:ib].cpu().numpy(), want))
    print("%s k=%d%s: %.3f ms per pass  %.3g k-mers/s  %.3g residues/s  (sketch stage %.3f ms, finalize %.3f ms)%s" %
          (algo, k, "" if algo == "hmh" else " p=%d" % p, dt * 1e3, kmers / dt, total / dt, tm["sketch_ms"] / steps, tm["finalize_ms"] / steps, ok))
ctx.close()
