#!/usr/bin/env python3
"""tests/leak_check_gpu.py — 150 context create / sketch / dist / destroy cycles; device memory must not creep.  Manual, GPU box."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, lash_amd
import oracle_lib as O
gs = [[O.synth_genome(i, 300_000).tobytes()] for i in range(8)] + [[b"ACGTN" * 1000]]
seq, off, goff = lash_amd.records_to_arrays(gs)
free0 = torch.cuda.mem_get_info()[0]
for it in range(150):
    c = lash_amd.Context(0)
    for an, k, p in (("hmh", 16, 0), ("hll", 21, 16), ("ull", 16, 12)):
        c.sketch_batch(an, k, p, 42, seq, off, goff)
    c.hmh_pair_counts(c.sketch_batch("hmh", 16, 0, 42, seq, off, goff), c.sketch_batch("hmh", 16, 0, 42, seq, off, goff))
    c.close()
    if it % 50 == 49:
        print(it + 1, "contexts: free memory delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6)
