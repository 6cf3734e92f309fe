#!/usr/bin/env python3
"""tools/dist_rate.py — pairs/s of the dist-side pair kernels on resident images (GPU box)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd
from lash_amd._lib import load
import ctypes as C

n = int(os.environ.get("N", 2048))
lib = load()
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
g = torch.Generator(device="cuda").manual_seed(1)
# HMH: random u16 registers from a small alphabet so that equal / zero cases occur
img = torch.randint(0, 7, (n, 16384), dtype=torch.int16, device="cuda", generator=g).view(torch.uint8).reshape(n, 32768).contiguous()
c = torch.zeros((n, n), dtype=torch.int32, device="cuda")
m = torch.zeros((n, n), dtype=torch.int32, device="cuda")
def run_hmh():
    rc = lib.lash_hmh_pair_counts_device(ctx._h, img.data_ptr(), n, img.data_ptr(), n, c.data_ptr(), m.data_ptr())
    assert rc == 0
run_hmh(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): run_hmh()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("hmh pairs: %d x %d in %.2f ms -> %.3g pairs/s (%.3g register pairs/s)" % (n, n, dt * 1e3, n * n / dt, n * n * 16384 / dt))
# check a few entries
a = img.view(torch.int16).reshape(n, 16384)
for (i, j) in ((0, 0), (3, 17), (n - 1, 5)):
    wc = int(((a[i] == a[j]) & (a[i] != 0)).sum()); wn = int(((a[i] != 0) | (a[j] != 0)).sum())
    assert int(c[i, j]) == wc and int(m[i, j]) == wn, (i, j)
for p, lo_, hi_, what in ((12, 0, 30, "uniform 0..29"), (14, 0, 30, "uniform 0..29"), (14, 7, 27, "genome-like band 7..26")):
    ib = 33 + (1 << p)
    h = torch.randint(lo_, hi_, (n, ib), dtype=torch.uint8, device="cuda", generator=g)
    z = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    s = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    def run_hll():
        rc = lib.lash_hll_pair_union_stats_device(ctx._h, p, h.data_ptr(), n, h.data_ptr(), n, z.data_ptr(), s.data_ptr())
        assert rc == 0
    run_hll(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run_hll()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("hll p=%d pairs (%s): %d x %d in %.2f ms -> %.3g pairs/s (%.3g register pairs/s)" % (p, what, n, n, dt * 1e3, n * n / dt, n * n * (1 << p) / dt))
# UltraLogLog: realistic register values (genome-like sketches: values within a band of ~24 around 4 * (p + log2(n/m)))
for p, est in ((12, "fgra"), (12, "ml"), (10, "fgra"), (16, "fgra")):
    ib = 8 + (1 << p)
    nn = n if p < 16 else max(64, n // 8)
    u = torch.randint(4 * p + 12, 4 * p + 44, (nn, ib), dtype=torch.uint8, device="cuda", generator=g)   # (p = 12: 60..91)
    e = torch.zeros((nn, nn), dtype=torch.float64, device="cuda")
    def run_ull():
        ctx.ull_pair_union_estimates_device(p, est, u, nn, u, nn, e)
    run_ull(); torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run_ull()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("ull p=%d %s pairs: %d x %d in %.2f ms -> %.3g pairs/s (%.3g register pairs/s)" % (p, est, nn, nn, dt * 1e3, nn * nn / dt, nn * nn * (1 << p) / dt))
# hyperminhash's expected-collision term for small sketches: cell vectors + f64 MFMA product
rc_, qc_ = np.random.default_rng(3).uniform(1e3, 5e5, n), np.random.default_rng(4).uniform(1e3, 5e5, n)
ctx.hmh_pair_expected_collisions(rc_, qc_[::-1].copy())          # (allocates the vector buffers; other queries than the timed call)
t0 = time.perf_counter()
ec = ctx.hmh_pair_expected_collisions(rc_, qc_)
dt = time.perf_counter() - t0
print("hmh expected collisions (both sketches < 2^19): %d x %d in %.1f ms -> %.3g pairs/s (%.2f TFLOP/s f64 incl. vectors and copy-back)"
      % (n, n, dt * 1e3, n * n / dt, n * n * 131072 / dt / 1e12))
print("ok")
