#!/usr/bin/env python3
"""tools/host_entry_rate.py — PCIe-inclusive rate of lash_sketch_batch (host buffers in, host images out) on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import lash_amd

G, L = int(os.environ.get("G", 200)), 5_000_000
_c = lash_amd.Context(0)
_d = torch.empty(G * L, dtype=torch.uint8, device="cuda")
_c.synth_genomes_device(0, G, L, _d)                      # the library's own generator
torch.cuda.synchronize()
seq = _d.cpu().numpy()
del _d
_c.close()
off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
goff = np.arange(G + 1, dtype=np.uint64)
ctx = lash_amd.Context(0)
for name, buf in (("pageable", seq), ):
    ctx.sketch_batch("hmh", 16, 0, 42, buf, off, goff)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        img = ctx.sketch_batch("hmh", 16, 0, 42, buf, off, goff)
    dt = (time.perf_counter() - t0) / n
    print("%s: %.1f ms per %d-genome batch  %.2f GB/s  %.3g k-mers/s" % (name, dt * 1e3, G, G * L / dt / 1e9, G * (L - 15) / dt))
img2 = ctx.sketch_batch("hmh", 16, 0, 42, seq, off, goff, flags=lash_amd.F_NO_DIRECT)
print("direct and pack-first routes agree:", np.array_equal(img, img2))
