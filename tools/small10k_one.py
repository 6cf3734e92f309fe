#!/usr/bin/env python3
"""tools/small10k_one.py — 100 000 x 10 kbp genomes, one sketch type (ALGO=hmh|hll), a few calls: for tools/pmc_py.sh (GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
G, L = 100_000, 10_000
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, G, L, d_seq)
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.arange(G + 1, dtype=np.uint64)
algo, k, p = os.environ.get("ALGO", "hmh"), 16, 10
if algo == "hll": k = 21
d_img = torch.zeros(G * lash_amd.image_bytes(algo, p), dtype=torch.uint8, device="cuda")
for _ in range(6):
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
torch.cuda.synchronize()
print("%s: %.3f ms per call" % (algo, (time.perf_counter() - t0) / 10 * 1e3))
