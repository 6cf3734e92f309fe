#!/usr/bin/env python3
"""tools/reads_n_probe.py — what isolated N bytes cost on one 3 Gbp genome cut into 150-bp reads vs the same bytes as one record (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, lash_amd
N, RL, k = 20_000_000, 150, 16
L = N * RL
dev = torch.device("cuda:0")
ctx0 = lash_amd.Context(0)
clean = torch.empty(L, dtype=torch.uint8, device=dev)
ctx0.synth_genomes_device(900000, 1, L, clean); ctx0.synchronize()
rng = np.random.default_rng(3)
d_seq = clean.clone()
rec_reads = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * RL
rec_one = torch.tensor([0, L], dtype=torch.int64, device=dev)
gbo = np.array([0, L], dtype=np.uint64)
for an, p in (("ull", 12), ("hmh", 0)):
    d_img = torch.zeros(lash_amd.image_bytes(an, p), dtype=torch.uint8, device=dev)
    for frac in (0.0, 0.0001, 0.001):
        d_seq.copy_(clean)
        if frac:
            idx = torch.from_numpy(rng.choice(N, size=int(N * frac), replace=False).astype(np.int64)).to(dev)
            d_seq[idx * RL + 37] = ord("N")
        torch.cuda.synchronize()
        for name, d_rec, nr in (("reads", rec_reads, N), ("one record", rec_one, 1)):
            c = lash_amd.Context(0)
            goff = np.array([0, nr], dtype=np.uint64)
            for _ in range(2):
                c.sketch_batch_device(an, k, p, 42, d_seq, d_rec, nr, goff, gbo, d_img)
            c.synchronize(); c.enable_timing(True)
            for _ in range(5):
                c.sketch_batch_device(an, k, p, 42, d_seq, d_rec, nr, goff, gbo, d_img)
            c.synchronize(); t = c.timing()
            print("%s %-10s N in %.4f of the reads (%6d): direct kernel %.3f ms, sketch stage %.3f ms, direct launches %d of 5" %
                  (an, name, frac, int(N * frac), t["direct_ms"] / max(t["direct_launches"], 1), t["sketch_ms"] / 5, t["direct_launches"]), flush=True)
            c.close()
