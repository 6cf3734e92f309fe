#!/usr/bin/env python3
"""tools/small_genomes_rate.py — sketch rate on many small genomes (viral / plasmid collections).  GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd

ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
for G, L in ((100_000, 10_000), (20_000, 50_000), (2_000, 500_000), (1_000_000, 1_000)):
    d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
    ctx.synth_genomes_device(0, G, L, d_seq)
    rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
    goff = np.arange(G + 1, dtype=np.uint64)
    for algo, k, p in (("hmh", 16, 0), ("hll", 21, 10)):
        ib = lash_amd.image_bytes(algo, p)
        d_img = torch.zeros(G * ib, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
        torch.cuda.synchronize()
        # the wall clock of back-to-back calls WITHOUT stage events (an event pair per stage costs the short calls 3-5 %) ...
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        # ... then three calls with them for the stage split
        ctx.enable_timing(True)
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t1) / 3
        tm = ctx.timing()
        ctx.enable_timing(False)
        print("%7d genomes x %7d bp %s: %7.2f ms  %.3g bases/s  | with stage events: %.2f ms = sketch %.2f + finalize %.2f + host and gaps %.2f" %
              (G, L, algo, dt * 1e3, G * L / dt, dte * 1e3, tm["sketch_ms"] / 3, tm["finalize_ms"] / 3,
               dte * 1e3 - (tm["sketch_ms"] + tm["finalize_ms"] + tm["pack_ms"]) / 3))
        del d_img
    del d_seq
