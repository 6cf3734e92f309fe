#!/usr/bin/env python3
"""tools/one_shape.py G L [algo k p] — a few sketch_batch_device calls on G synthetic genomes of L bases (for rocprofv3)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
G, L = int(sys.argv[1]), int(sys.argv[2])
algo, k, p = (sys.argv[3], int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ("hmh", 16, 0)
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, G, L, d_seq)
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.arange(G + 1, dtype=np.uint64)
d_img = torch.zeros(G * lash_amd.image_bytes(algo, p), dtype=torch.uint8, device="cuda")
for _ in range(4):
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
torch.cuda.synchronize()
