#!/usr/bin/env python3
"""tools/varreads_rate.py — one sketch of N reads of UNEQUAL length (trimmed FASTQ: 100..150 bp) vs equal length (150 bp), ull p=12 k=16
(GPU box): the record-start bitmap route against the arithmetic one."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lash_amd

N, k = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 16
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
for name, lens in (("equal 150 bp", np.full(N, 150, np.int64)), ("100..150 bp", rng.integers(100, 151, size=N).astype(np.int64))):
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    L = int(off[-1])
    ctx = lash_amd.Context(0)
    d_seq = torch.empty(L, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(900000, 1, L, d_seq)
    if os.environ.get("N_FRAC"):                               # an N in that fraction of the reads (what real FASTQ has)
        idx = torch.from_numpy(rng.choice(N, size=int(N * float(os.environ["N_FRAC"])), replace=False).astype(np.int64)).to(dev)
        at = torch.from_numpy(off[:-1].astype(np.int64)).to(dev)[idx] + 37
        d_seq[at] = ord("N")
    d_rec = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_img = torch.zeros(lash_amd.image_bytes("ull", 12), dtype=torch.uint8, device=dev)
    goff = np.array([0, N], dtype=np.uint64)
    gbo = np.array([0, L], dtype=np.uint64)
    for _ in range(3):
        ctx.sketch_batch_device("ull", k, 12, 42, d_seq, d_rec, N, goff, gbo, d_img)
    ctx.synchronize()
    ctx.enable_timing(True)
    for _ in range(10):
        ctx.sketch_batch_device("ull", k, 12, 42, d_seq, d_rec, N, goff, gbo, d_img)
    ctx.synchronize()
    t = ctx.timing()
    ms = (t["sketch_ms"] + t["finalize_ms"] + t["pack_ms"]) / 10
    print("%-14s %d reads, %.2f GB: %.3f ms per step -> %.4g k-mers/s (%.4g B/s), direct launches %d of 10" % (name, N, L / 1e9, ms, t["kmers"] / 10 / (ms * 1e-3), L / (ms * 1e-3), t["direct_launches"]), flush=True)
    ctx.close()
    del d_seq, d_rec
