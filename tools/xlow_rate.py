#!/usr/bin/env python3
"""tools/xlow_rate.py — HyperMinHash rate with LASH_F_HMH_X_LOW (switch U1: x = low half of xxh3_128) vs the default."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
G, L = 400, 5_000_000
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, G, L, d_seq)
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.arange(G + 1, dtype=np.uint64)
d_img = torch.zeros(G * 32768, dtype=torch.uint8, device="cuda")
for flags, name in ((0, "x = high64 (default)"), (lash_amd.F_HMH_X_LOW, "x = low64 (LASH_F_HMH_X_LOW)")):
    for k in (16, 31):
        for _ in range(4):
            ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img, flags=flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img, flags=flags)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 6
        print("%-30s k=%d: %.3f ms  %.4g k-mers/s" % (name, k, dt * 1e3, G * (L - k + 1) / dt))
