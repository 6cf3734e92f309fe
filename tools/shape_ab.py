#!/usr/bin/env python3
"""tools/shape_ab.py G L algo k p [reps] — one shape, wall clock per call and the launch statistics (for A/B runs under environment switches)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
G, L, algo, k, p = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, G, L, d_seq)
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.arange(G + 1, dtype=np.uint64)
d_img = torch.zeros(G * lash_amd.image_bytes(algo, p), dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
ctx.enable_timing(True)
ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
torch.cuda.synchronize()
tm = ctx.timing()
print("%s %d x %d %s k=%d p=%d: %.3f ms per call; %s" % (" ".join("%s=%s" % (e, os.environ[e]) for e in sorted(os.environ) if e.startswith("LASH_")), G, L, algo, k, p, dt * 1e3,
      {x: (round(v, 3) if isinstance(v, float) else v) for x, v in tm.items()}), flush=True)
