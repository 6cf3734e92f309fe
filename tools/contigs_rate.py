#!/usr/bin/env python3
"""tools/contigs_rate.py — soft-masked multi-contig assemblies through the sketch stage (GPU box): G genomes of 5 Mbp cut into 50
contigs of unequal length, every other 10 kb lower-case; default route and LASH_F_STREAM_ONLY."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lash_amd

G, L, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 5_000_000, 16
dev = torch.device("cuda:0")
ctx = lash_amd.Context(0)
d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
ctx.synth_genomes_device(0, G, L, d_seq)
ctx.synchronize()
rng = np.random.default_rng(5)
cuts = np.sort(rng.choice(np.arange(1000, L - 1000), size=49, replace=False))
per = np.concatenate([[0], cuts, [L]]).astype(np.uint64)
rec_off = (np.arange(G, dtype=np.uint64)[:, None] * np.uint64(L) + per[None, :-1]).reshape(-1)
rec_off = np.concatenate([rec_off, [np.uint64(G * L)]])
goff = np.arange(G + 1, dtype=np.uint64) * np.uint64(50)
gbo = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
d_img = torch.zeros(G * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device=dev)
for masked in (False, True):
    if masked:
        pos = torch.arange(L, device=dev) % 20000
        d_seq.view(G, L)[:, pos >= 10000] |= 0x20
        torch.cuda.synchronize()
    for name, flags in (("default route", 0), ("stream only", lash_amd.F_STREAM_ONLY)):
        c = lash_amd.Context(0)
        for _ in range(3):
            c.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, len(rec_off) - 1, goff, gbo, d_img, flags=flags)
        c.synchronize()
        c.enable_timing(True)
        for _ in range(10):
            c.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, len(rec_off) - 1, goff, gbo, d_img, flags=flags)
        c.synchronize()
        t = c.timing()
        ms = (t["sketch_ms"] + t["finalize_ms"] + t["pack_ms"]) / 10
        print("%-9s %-14s %.3f ms per %d x 5 Mbp in 50 contigs -> %.4g k-mers/s" % ("masked" if masked else "clean", name, ms, G, t["kmers"] / 10 / (ms * 1e-3)), flush=True)
        c.close()
