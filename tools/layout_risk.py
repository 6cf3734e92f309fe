#!/usr/bin/env python3
"""tools/layout_risk.py [genomes] — what a flipped layout switch costs.

The crate-internal rules the reference's sketches depend on are guesses until tools/ref_probe/ runs: every one of
them is a field of lash_layout.  This tool sketches the three judged shapes (hmh k=16, hll p=14 k=21, ull p=12 k=16) on G x 5 Mbp under the
default and under every SINGLE-switch alternative, from ASCII (lash_sketch_batch_device: what `lash sketch` calls) and from resident
2-bit genomes (lash_sketch_packed_device), and prints each rate with its ratio to the default's.  Each alternative's images are
also compared between the two routes (two kernel chains) — the oracle comparison is tests/test_gpu_layout.py's."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lash_amd

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = 5_000_000
LAYOUTS = [("default", None), ("hmh_x=low", "hmh_x=low"), ("kmer=lsb", "kmer=lsb"), ("codes=ACTG", "codes=ACTG"), ("codes=TGCA", "codes=TGCA"),
           ("hll_bucket=high", "hll_bucket=high"), ("hmh_x=low,kmer=lsb", "hmh_x=low,kmer=lsb")]
SHAPES = [("hmh", 16, 0), ("hll", 21, 14), ("ull", 16, 12)]
APPLIES = {"hmh_x=low": ("hmh",), "hll_bucket=high": ("hll",), "hmh_x=low,kmer=lsb": ("hmh",)}

ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, G, L, d_seq)
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.arange(G + 1, dtype=np.uint64)


def timed(fn, reps=8, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tm = ctx.timing()
    ctx.enable_timing(False)
    return dt, tm


print("layout risk: %d x %d bp; rates in k-mers/s (wall clock over the whole call; sketch stage by HIP events in brackets)" % (G, L))
base = {}
for name, spec in LAYOUTS:
    ctx.set_layout(spec)
    for algo, k, p in SHAPES:
        if name in APPLIES and algo not in APPLIES[name]:
            continue
        ib = ctx.image_bytes(algo, p)
        nk = G * (L - k + 1)
        d_a = torch.zeros(G * ib, dtype=torch.uint8, device="cuda")
        d_b = torch.zeros(G * ib, dtype=torch.uint8, device="cuda")
        dt_d, tm_d = timed(lambda: ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_a))
        pk = ctx.pack_device(d_seq, d_rec, G, goff, rec_off)
        dt_p, tm_p = timed(lambda: ctx.sketch_packed_device(algo, k, p, 42, pk, d_b))
        pk.free()
        same = bool(torch.equal(d_a, d_b))
        key = (algo, k, p)
        if name == "default":
            base[key] = (dt_d, dt_p)
        print("%-20s %s k=%-2d p=%-2d  ascii %7.3f ms %.4g (x%.2f of default) [sketch %.3f ms, direct launches %d, defer %d]   packed %7.3f ms %.4g (x%.2f) [sketch %.3f ms]  routes %s"
              % (name, algo, k, p, dt_d * 1e3, nk / dt_d, base[key][0] / dt_d, tm_d["sketch_ms"] / max(tm_d["calls"], 1), tm_d["direct_launches"], tm_d["defer_launches"],
                 dt_p * 1e3, nk / dt_p, base[key][1] / dt_p, tm_p["sketch_ms"] / max(tm_p["calls"], 1), "agree" if same else "DIFFER"), flush=True)
        del d_a, d_b
ctx.set_layout(None)
