#!/usr/bin/env python3
"""tools/host_entry_rate.py — PCIe-inclusive rate of the host-buffer entries on the GPU box:
lash_sketch_batch (synchronous, one batch at a time) from pageable and from page-locked memory, and the streaming loop a
host is meant to write — lash_sketch_batch_async over two pinned buffers, the H2D copy of batch n+1 overlapping the kernels
of batch n — next to the bare H2D copy rate of the same buffers (the ceiling of any host-buffer entry)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import lash_amd

G, L, NB = int(os.environ.get("G", 200)), 5_000_000, int(os.environ.get("BATCHES", 8))
ctx = lash_amd.Context(0)
d = torch.empty(G * L, dtype=torch.uint8, device="cuda")
off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
goff = np.arange(G + 1, dtype=np.uint64)
ib = lash_amd.image_bytes("hmh")
pins = [lash_amd.PinnedArray(G * L) for _ in range(2)]
offs = [lash_amd.PinnedArray((G + 1) * 8, np.uint64) for _ in range(2)]
outs = [lash_amd.PinnedArray(G * ib) for _ in range(2)]
for i in range(2):
    ctx.synth_genomes_device(1000 * i, G, L, d)                # the library's own generator
    torch.cuda.synchronize()
    ctx.synchronize()
    pins[i].array[:] = d.cpu().numpy()
    offs[i].array[:] = off
pageable = pins[0].array.copy()

# the ceiling: bare H2D of the pinned buffer
t = torch.empty(G * L, dtype=torch.uint8, device="cuda")
src = torch.from_numpy(pins[0].array)
for _ in range(2):
    t.copy_(src, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    t.copy_(src, non_blocking=True)
torch.cuda.synchronize()
h2d = 5 * G * L / (time.perf_counter() - t0)
print("bare H2D copy of one pinned batch: %.1f GB/s" % (h2d / 1e9))

for name, buf in (("pageable", pageable), ("pinned  ", pins[0].array)):
    ctx.sketch_batch("hmh", 16, 0, 42, buf, off, goff)
    t0 = time.perf_counter()
    for _ in range(3):
        img = ctx.sketch_batch("hmh", 16, 0, 42, buf, off, goff)
    dt = (time.perf_counter() - t0) / 3
    print("lash_sketch_batch, %s: %.1f ms per %d-genome batch  %.2f GB/s  %.3g k-mers/s" % (name, dt * 1e3, G, G * L / dt / 1e9, G * (L - 15) / dt))
want = [ctx.sketch_batch("hmh", 16, 0, 42, pins[i].array, off, goff) for i in range(2)]

for _ in range(2):                                            # warm the slots
    for i in range(2):
        ctx.sketch_batch_async("hmh", 16, 0, 42, pins[i].array, offs[i].array, goff, outs[i].array)
ctx.synchronize()
t0 = time.perf_counter()
for b in range(NB):
    i = b & 1
    ctx.sketch_batch_async("hmh", 16, 0, 42, pins[i].array, offs[i].array, goff, outs[i].array)
ctx.synchronize()
dt = (time.perf_counter() - t0) / NB
print("lash_sketch_batch_async over two pinned buffers: %.1f ms per batch  %.2f GB/s  %.3g k-mers/s  = %.0f %% of the bare H2D rate"
      % (dt * 1e3, G * L / dt / 1e9, G * (L - 15) / dt, 100 * (G * L / dt) / h2d))
ok = all(np.array_equal(outs[i].array.reshape(G, ib), want[i]) for i in range(2))
print("async images identical to the synchronous entry:", ok)
sys.exit(0 if ok else 1)
