#!/usr/bin/env python3
"""tools/sparse_n_rate.py — isolated N bytes at a fixed spacing in 1 000 x 5 Mbp single-record genomes (GPU box): what a sparsely
dirty wave-tile costs on the direct pass (junction walks)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lash_amd

G, L, k = 1000, 5_000_000, 16
dev = torch.device("cuda:0")
ctx0 = lash_amd.Context(0)
clean = torch.empty(G * L, dtype=torch.uint8, device=dev)
ctx0.synth_genomes_device(0, G, L, clean)
ctx0.synchronize()
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
goff = np.arange(G + 1, dtype=np.uint64)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
d_img = torch.zeros(G * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device=dev)
d_seq = clean.clone()
for spacing in ([int(x) for x in os.environ["SPACINGS"].split(",")] if os.environ.get("SPACINGS") else (0, 1_000_000, 100_000, 30_000, 10_000)):
    d_seq.copy_(clean)
    if spacing:
        d_seq.view(G, L)[:, 777::spacing] = ord("N")
    torch.cuda.synchronize()
    c = lash_amd.Context(0)
    for _ in range(3):
        c.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    c.synchronize()
    c.enable_timing(True)
    for _ in range(10):
        c.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    c.synchronize()
    t = c.timing()
    ms = (t["sketch_ms"] + t["finalize_ms"]) / 10
    n_dirty = (L // spacing + 1) * G if spacing else 0
    print("one N every %9d bytes (%7d dirty wave-tiles of %d): %.3f ms per step, direct launches %d of 10" % (spacing, n_dirty, G * L // 4096, ms, t["direct_launches"]), flush=True)
    c.close()
