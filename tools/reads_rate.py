#!/usr/bin/env python3
"""tools/reads_rate.py — sketch rate on short-read shaped input (BASELINE configs[4] shape): one input of N x 150-bp records
resident in HBM (records mode), k-mers never span records.  GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd

n_reads, L = int(os.environ.get("READS", 6_666_667)), 150
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, 1, n_reads * L, d_seq)
rec_off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.array([0, n_reads], dtype=np.uint64)
gbo = np.array([0, n_reads * L], dtype=np.uint64)
for algo, k, p in (("hmh", 16, 0), ("ull", 16, 12), ("hll", 21, 14)):
    ib = lash_amd.image_bytes(algo, p)
    d_img = torch.zeros(ib, dtype=torch.uint8, device="cuda")
    for flags, name in ((0, "direct"), (lash_amd.F_NO_DIRECT, "pack-first")):
        for _ in range(3):
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, n_reads, goff, gbo, d_img, flags=flags)
        torch.cuda.synchronize()
        ctx.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, n_reads, goff, gbo, d_img, flags=flags)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        tm = ctx.timing()
        ctx.enable_timing(False)
        kmers = n_reads * (L - k + 1)
        assert tm["kmers"] == kmers * 5
        print("%s k=%d %-10s: %.2f ms per Gbp-batch  %.3g k-mers/s  %.3g bases/s  stages pack %.2f sketch %.2f finalize %.2f" %
              (algo, k, name, dt * 1e3, kmers / dt, n_reads * L / dt, tm["pack_ms"] / 5, tm["sketch_ms"] / 5, tm["finalize_ms"] / 5))
