#!/usr/bin/env python3
"""tools/dirty_rate.py [genomes] — the direct sketch pass on soft-masked input: time per step for lower-case blocks of
several sizes and densities (GPU box).  `frac` of every period of 2*B bytes is lower-case.  Prints the kernel time from the
context's own events (lash_timing.direct_ms / sketch_ms), the k-mers of the step and the rate."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lash_amd


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    L, k = 5_000_000, 16
    dev = torch.device("cuda:0")
    ctx = lash_amd.Context(0)
    d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(0, G, L, d_seq)
    ctx.synchronize()
    clean = d_seq.clone()
    rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    goff = np.arange(G + 1, dtype=np.uint64)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
    d_img = torch.zeros(G * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device=dev)
    cases = [("clean", 0, 0)] + [("B=%d lower %d%%" % (B, int(100 * lo / (2 * B))), B, lo)
                                  for B, lo in ((500, 500), (2_500, 2_500), (10_000, 10_000), (100_000, 100_000), (2_500_000, 2_500_000),
                                                (10_000, 2_000), (10_000, 18_000), (500, 100),
                                                # block edges at arbitrary byte positions (the ones above all fall on multiples of 4 or 16 bytes)
                                                (10_007, 10_007), (2_503, 2_503), (509, 509))] + \
            [("random blocks 30..30000 B", -2, 0), ("all lower", -1, 0)]
    for name, B, lo in cases:
        d_seq.copy_(clean)
        v = d_seq.view(G, L)
        if B == -1:
            d_seq |= 0x20
        elif B == -2:
            # a RepeatMasker-like mask: alternating upper / lower runs of log-uniform length, the same for every genome
            rng = np.random.default_rng(5)
            edges = np.cumsum(np.exp(rng.uniform(np.log(30), np.log(30000), size=4 * L // 3000)).astype(np.int64))
            edges = edges[edges < L]
            lower = np.zeros(L + 1, np.int8)
            lower[edges[0::2]] += 1
            lower[edges[1::2]] -= 1
            mask = torch.from_numpy(np.cumsum(lower[:L]).astype(np.bool_)).to(dev)
            v[:, mask] |= 0x20
            del mask
        elif B:
            per = 2 * B
            pos = torch.arange(L, device=dev) % per
            v[:, pos >= per - lo] |= 0x20
        torch.cuda.synchronize()
        ctx.close()
        ctx = lash_amd.Context(0)                      # a fresh context: the direct pass's back-off state belongs to the context
        ctx.enable_timing(True)
        for _ in range(3):
            ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
        ctx.synchronize()
        ctx.enable_timing(True)
        steps = 10
        for _ in range(steps):
            ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
        ctx.synchronize()
        t = ctx.timing()
        ms = (t["direct_ms"] if t.get("direct_ms") else t["sketch_ms"]) / steps
        tot = (t["pack_ms"] + t["sketch_ms"] + t["finalize_ms"]) / steps
        kmers = t["kmers"] / steps
        print("%-28s direct %.3f ms  pack+sketch+finalize %.3f ms  k-mers/step %.4g  -> %.4g k-mers/s (bytes %.4g B/s)  direct launches %d"
              % (name, ms, tot, kmers, kmers / (tot * 1e-3), G * L / (tot * 1e-3), t.get("direct_launches", -1)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
