#!/usr/bin/env python3
"""tools/shape_seq_trace.py — LASH_TRACE_HOST=1: host marks of ONE call on 2 000 x 500 kbp, in a fresh context and after a call on 100 000 x 10 kbp."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
def shape(G, L, reps, say):
    d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
    ctx.synth_genomes_device(0, G, L, d_seq)
    rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
    goff = np.arange(G + 1, dtype=np.uint64)
    d_img = torch.zeros(G * 32768, dtype=torch.uint8, device="cuda")
    for i in range(reps):
        if i == reps - 1 and say:
            torch.cuda.synchronize(); print("---- traced call:", say, file=sys.stderr, flush=True)
        ctx.sketch_batch_device("hmh", 16, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    torch.cuda.synchronize()
os.environ.pop("X", None)
shape(2000, 500000, 4, "2000 x 500k, fresh context")
shape(100000, 10000, 3, None)
shape(2000, 500000, 4, "2000 x 500k after 100000 x 10k")
