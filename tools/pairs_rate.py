#!/usr/bin/env python3
"""tools/pairs_rate.py — pairs/s of the resident all-vs-all (lash_sketch_set_pair_block_device) on N synthetic sketches (GPU box).

    N=16384 ALGO=hmh python tools/pairs_rate.py        # triangle of N x N in row blocks, device outputs, HIP-synchronised wall time
LASH_HMH_PAIRS_WORDS=1 selects the u16-pair kernel for the A/B."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
from lash_amd._lib import load

n = int(os.environ.get("N", 16384))
algo = os.environ.get("ALGO", "hmh")
p = int(os.environ.get("P", 14 if algo == "hll" else 12))
rows = int(os.environ.get("BLOCK_ROWS", 2048))
full = os.environ.get("FULL", "1") == "1"
lib = load()
ctx = lash_amd.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
if algo == "hmh":
    p = 0
    # genome-like: no zero register (FULL=0: a small alphabet with zeros, the general form)
    img = (torch.randint(1 if full else 0, 60000 if full else 7, (n, 16384), dtype=torch.int32, device="cuda", generator=g).to(torch.int16)
           .view(torch.uint8).reshape(n, 32768).contiguous())
elif algo == "hll":
    img = torch.randint(7, 27, (n, 33 + (1 << p)), dtype=torch.uint8, device="cuda", generator=g)
else:
    img = torch.randint(4 * p + 12, 4 * p + 44, (n, 8 + (1 << p)), dtype=torch.uint8, device="cuda", generator=g)
torch.cuda.synchronize()
t0 = time.perf_counter()
s = ctx.sketch_set(algo, p, img)
s.prepare()
ctx.synchronize()
t_prep = time.perf_counter() - t0
t0 = time.perf_counter()
card = s.cardinalities()
t_card = time.perf_counter() - t0
c = torch.empty(rows * n, dtype=torch.int32, device="cuda")
m = torch.empty(rows * n, dtype=torch.int32, device="cuda")
u = torch.empty(rows * n, dtype=torch.float64, device="cuda")


def sweep(triangle):
    pairs = 0
    for r0 in range(0, n, rows):
        r1 = min(n, r0 + rows)
        nc = r1 if triangle else n
        rc = lib.lash_sketch_set_pair_block_device(ctx._h, s._h, r0, r1, s._h, nc, 1 if triangle else 0, 0, c.data_ptr(), m.data_ptr(), u.data_ptr())
        assert rc == 0, rc
        pairs += sum(min(i + 1, nc) for i in range(r0, r1)) if triangle else (r1 - r0) * nc
    ctx.synchronize()
    return pairs


sweep(True)
for tri in (True, False):
    t0 = time.perf_counter()
    pairs = sweep(tri)
    dt = time.perf_counter() - t0
    regs = 16384 if algo == "hmh" else (1 << p)
    print("%s p=%d %s: N=%d, %.4g wanted pairs in %.1f ms -> %.3g pairs/s (%.3g register pairs/s); prepare %.1f ms, cardinalities %.1f ms"
          % (algo, p, "triangle" if tri else "full", n, pairs, dt * 1e3, pairs / dt, pairs * regs / dt, t_prep * 1e3, t_card * 1e3))
# spot check against torch
if algo == "hmh":
    a = img.view(torch.int16).reshape(n, 16384)
    r0 = (n - 1) // rows * rows
    lib.lash_sketch_set_pair_block_device(ctx._h, s._h, r0, n, s._h, n, 1, 0, c.data_ptr(), m.data_ptr(), u.data_ptr())
    ctx.synchronize()
    cc = c[: (n - r0) * n].view(n - r0, n)
    mm = m[: (n - r0) * n].view(n - r0, n)
    for (i, j) in ((n - 1, 0), (n - 1, n - 1), (r0, 17), (r0 + 5, r0 + 5)):
        wc = int(((a[i] == a[j]) & (a[i] != 0)).sum())
        wn = int(((a[i] != 0) | (a[j] != 0)).sum())
        assert int(cc[i - r0, j]) == wc and int(mm[i - r0, j]) == wn, (i, j, int(cc[i - r0, j]), wc)
print("ok")
