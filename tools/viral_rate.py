#!/usr/bin/env python3
"""tools/viral_rate.py [genomes] — very many small genomes of unequal size (a viral collection: 3..300 kbp, log-uniform, 1..4 records
each), ASCII resident in HBM, hmh k=16 and hll p=10 k=21 (GPU box): wall time per call (host planning included) and device stage times."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lash_amd

G = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(13)
dev = torch.device("cuda:0")
lens = np.exp(rng.uniform(np.log(3e3), np.log(3e5), size=G)).astype(np.int64)
gbo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
total = int(gbo[-1])
ctx = lash_amd.Context(0)
d_seq = torch.empty(total, dtype=torch.uint8, device=dev)
ctx.synth_genomes_device(0, 1, total, d_seq)
ctx.synchronize()
nrec = rng.integers(1, 5, size=G)
rec = []
for g in range(G):
    L, n = int(lens[g]), int(nrec[g])
    cuts = np.sort(rng.integers(1, L, size=n - 1)) if n > 1 else []
    rec.append(int(gbo[g]))
    rec.extend(int(gbo[g]) + int(c) for c in cuts)
rec.append(total)
rec_off = np.asarray(rec, dtype=np.uint64)
goff = np.concatenate([[0], np.cumsum(nrec)]).astype(np.uint64)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
print("%d genomes, %.2f GB, %d records" % (G, total / 1e9, len(rec_off) - 1), flush=True)
for an, k, p in (("hmh", 16, 0), ("hll", 21, 10)):
    d_img = torch.zeros(G * lash_amd.image_bytes(an, p), dtype=torch.uint8, device=dev)
    for _ in range(2):
        ctx.sketch_batch_device(an, k, p, 42, d_seq, d_rec, len(rec_off) - 1, goff, gbo, d_img)
    ctx.synchronize()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.sketch_batch_device(an, k, p, 42, d_seq, d_rec, len(rec_off) - 1, goff, gbo, d_img)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / 5
    t = ctx.timing()
    ctx.enable_timing(False)
    print("%s k=%d p=%d: wall %.2f ms per call; device: sketch stage %.2f ms, finalize %.2f ms -> %.4g k-mers/s on the wall clock" %
          (an, k, p, wall * 1e3, t["sketch_ms"] / 5, t["finalize_ms"] / 5, t["kmers"] / 5 / wall), flush=True)
