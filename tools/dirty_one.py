#!/usr/bin/env python3
"""tools/dirty_one.py B lo [genomes] — one soft-masked shape (period 2*B bytes, the last `lo` of them lower-case; B = 0: clean), a few calls:
for rocprofv3 (tools/profile_py.sh).  Prints the surviving k-mers per call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import lash_amd
B, lo = int(sys.argv[1]), int(sys.argv[2])
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
L, k = 5_000_000, 16
dev = torch.device("cuda:0")
ctx = lash_amd.Context(0)
d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
ctx.synth_genomes_device(0, G, L, d_seq)
ctx.synchronize()
if B:
    pos = torch.arange(L, device=dev) % (2 * B)
    d_seq.view(G, L)[:, pos >= 2 * B - lo] |= 0x20
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
goff = np.arange(G + 1, dtype=np.uint64)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
d_img = torch.zeros(G * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device=dev)
N = int(os.environ.get("DIRTY_ONE_CALLS", "6"))
for _ in range(4):                                     # the optimistic direct pass backs off after the first calls
    ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
ctx.synchronize()
ctx.enable_timing(True)
for _ in range(N):
    ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
ctx.synchronize()
t = ctx.timing()
print("B=%d lo=%d: %d k-mers per call, sketch stage %.3f ms per call (direct launches %d of %d calls)" % (B, lo, t["kmers"] // N, t["sketch_ms"] / N, t.get("direct_launches", -1), N))
