#!/usr/bin/env python3
"""tools/shape_seq.py "G L algo k p" ... — several shapes one after the other on ONE context (does a launch depend on what ran before it?)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
for spec in sys.argv[1:]:
    G, L, algo, k, p = spec.split()
    G, L, k, p = int(G), int(L), int(k), int(p)
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
    for _ in range(10):
        ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    ctx.enable_timing(True)
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    torch.cuda.synchronize()
    tm = ctx.timing()
    ctx.enable_timing(False)
    print("%s: %.3f ms per call; %s" % (spec, dt * 1e3, {x: (round(v, 3) if isinstance(v, float) else v) for x, v in tm.items()}), flush=True)
    del d_seq, d_img
