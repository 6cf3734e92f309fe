#!/usr/bin/env python3
"""tools/find_hll_corner.py — find k-mers whose HyperLogLog rank exceeds 53 - p (GPU box).

streaming_algorithms keeps `sum = sum_j 2^-m[j]` incrementally in f64; every update is exact while all
registers stay <= 53 - p, and order-dependent rounding starts above that.  Such a k-mer turns up once per 2^(52-p) hashes
(p = 16: 1.4e11), so the only practical way to get a test vector is to look for one with the sketch kernel itself:
sketch batches of synthetic genomes, spot a register above the bound, bisect the genome down to the 21-mer.
Prints one line per find: p k seed rho bucket kmer."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd

p = int(os.environ.get("P", 16))
k = int(os.environ.get("K", 21))
seed = int(os.environ.get("SEED", 42))
batches = int(os.environ.get("BATCHES", 200))
want = int(os.environ.get("FINDS", 4))
G, L = 1000, 5_000_000
bound = 53 - p

ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
ib = ctx.image_bytes("hll", p)
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
d_rec = torch.arange(G + 1, dtype=torch.int64, device="cuda") * L
d_img = torch.empty((G, ib), dtype=torch.uint8, device="cuda")
goff = np.arange(G + 1, dtype=np.uint64)
gbo = goff * np.uint64(L)
one_img = torch.empty((1, ib), dtype=torch.uint8, device="cuda")
one_rec = torch.zeros(2, dtype=torch.int64, device="cuda")


def max_reg_of_slice(d_bytes):
    one_rec[1] = d_bytes.numel()
    ctx.sketch_batch_device("hll", k, p, seed, d_bytes, one_rec, 1, np.array([0, 1], np.uint64), np.array([0, d_bytes.numel()], np.uint64), one_img)
    torch.cuda.synchronize()
    return int(one_img[0, 33:].max())


finds = 0
for b in range(batches):
    ctx.synth_genomes_device(b * G, G, L, d_seq)
    ctx.sketch_batch_device("hll", k, p, seed, d_seq, d_rec, G, goff, gbo, d_img)
    torch.cuda.synchronize()
    mx = d_img[:, 33:].max(dim=1).values
    hot = torch.nonzero(mx > bound).flatten().tolist()
    for g in hot:
        rho = int(mx[g])
        bucket = int(torch.argmax(d_img[g, 33:]))
        seq = d_seq[g * L:(g + 1) * L]
        lo, hi = 0, L - k + 1                       # k-mer start positions [lo, hi)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if max_reg_of_slice(seq[lo:mid + k - 1].contiguous()) >= rho:
                hi = mid
            else:
                lo = mid
        kmer = bytes(seq[lo:lo + k].cpu().numpy().tobytes()).decode()
        assert max_reg_of_slice(seq[lo:lo + k].contiguous()) == rho
        print("FOUND p=%d k=%d seed=%d rho=%d bucket=%d genome=%d pos=%d kmer=%s" % (p, k, seed, rho, bucket, b * G + g, lo, kmer), flush=True)
        finds += 1
    if finds >= want:
        break
print("done: %d batches, %d finds" % (b + 1, finds))
