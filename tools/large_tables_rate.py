#!/usr/bin/env python3
"""tools/large_tables_rate.py — sketch rate with register tables beyond 128 KiB of LDS: N x 5 Mbp synthetic genomes
resident in HBM, hll p=16 k=21 and ull p=15..22 k=16.  Run twice for the A/B: as is (binned: hash once, scatter, one LDS pass per bin)
and with LASH_NO_BINS=1 (rounds 1-3: one global atomic per k-mer; before round 4 p <= 18 re-hashed per 128 KiB part).  GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd

G, L = int(os.environ.get("G", 1000)), 5_000_000
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, G, L, d_seq)
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.arange(G + 1, dtype=np.uint64)
shapes = [("hll", 21, 16)] + [("ull", 16, p) for p in (15, 16, 18, 20, 22)]
if os.environ.get("SHAPES"):
    shapes = [(a, int(k), int(p)) for a, k, p in (x.split(":") for x in os.environ["SHAPES"].split(","))]
print("%d x %d bp%s" % (G, L, "  LASH_NO_BINS=1" if os.environ.get("LASH_NO_BINS") else ""))
for algo, k, p in shapes:
    ib = lash_amd.image_bytes(algo, p)
    d_img = torch.zeros(G * ib, dtype=torch.uint8, device="cuda")
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    torch.cuda.synchronize()
    steps = 3
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = ctx.timing()
    ctx.enable_timing(False)
    kmers = G * (L - k + 1)
    assert tm["kmers"] == kmers * steps
    print("%s k=%d p=%d: %.2f ms per step  %.3g k-mers/s  (sketch stage %.2f ms, finalize %.2f ms)  sha %s" %
          (algo, k, p, dt * 1e3, kmers / dt, tm["sketch_ms"] / steps, tm["finalize_ms"] / steps, hex(hash(d_img[:4 * ib].cpu().numpy().tobytes()) & 0xFFFFFFFF)))
ctx.close()
