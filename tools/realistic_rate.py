#!/usr/bin/env python3
"""tools/realistic_rate.py [genomes] — a collection shaped like a public genome database rather than like the benchmark (GPU box):
genome lengths 0.6..12 Mbp (log-uniform), 1..400 contigs each of unequal length, 3 % of the genomes with N gaps between some contigs,
5 % soft-masked (a lower-case stretch of 200..3 000 bases every 1..8 kb), the rest clean; hmh k=16, hll p=14 k=21.  Prints the step time
(sketch stage + finalize) and k-mers/s beside the same bytes as ONE clean record per genome."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lash_amd

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(11)
dev = torch.device("cuda:0")
lens = np.exp(rng.uniform(np.log(6e5), np.log(1.2e7), size=G)).astype(np.int64)
gbo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
total = int(gbo[-1])
ctx = lash_amd.Context(0)
d_seq = torch.empty(total, dtype=torch.uint8, device=dev)
# one synthetic stream cut into genomes (the generator makes equal-length genomes; content is i.i.d. either way)
ctx.synth_genomes_device(0, 1, total, d_seq)
ctx.synchronize()
rec_off, goff = [0], [0]
kinds = rng.random(G)
if os.environ.get("ALL_CLEAN"):
    kinds[:] = 1.0                                         # (the same lengths and contigs without any dirt: what the mix itself costs)
for g in range(G):
    L = int(lens[g])
    n_contigs = int(min(400, max(1, rng.geometric(1 / 60.0)))) if rng.random() < 0.8 else 1
    cuts = np.sort(rng.choice(np.arange(1, L), size=min(n_contigs - 1, L - 1), replace=False)) if n_contigs > 1 else np.array([], np.int64)
    starts = np.concatenate([[0], cuts]) + int(gbo[g])
    rec_off.extend((starts[1:]).tolist() if len(starts) > 1 else [])
    rec_off.append(int(gbo[g + 1]))
    goff.append(len(rec_off) - 1)
    view = d_seq[int(gbo[g]):int(gbo[g + 1])]
    if kinds[g] < 0.03 and len(cuts):                          # N gaps at some contig ends
        for c in cuts[: max(1, len(cuts) // 5)]:
            n = int(rng.choice([10, 100, 1000, 20000]))
            view[max(0, int(c) - n):int(c)] = ord("N")
    elif kinds[g] < 0.08:                                      # soft-masked
        per = int(rng.integers(1000, 8000)); run = int(rng.integers(200, min(3000, per)))
        pos = torch.arange(L, device=dev) % per
        view[pos < run] |= 0x20
rec_off = np.asarray(rec_off, dtype=np.uint64)
goff = np.asarray(goff, dtype=np.uint64)
n_rec = len(rec_off) - 1
d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
d_rec1 = torch.from_numpy(gbo.astype(np.int64)).to(dev)
print("%d genomes, %.2f GB, %d records (%.0f per genome, mean %.0f bytes)" % (G, total / 1e9, n_rec, n_rec / G, total / n_rec), flush=True)
for an, k, p in (("hmh", 16, 0), ("hll", 21, 14)):
    d_img = torch.zeros(G * lash_amd.image_bytes(an, p), dtype=torch.uint8, device=dev)
    for name, dr, nr, go in (("as a database has them", d_rec, n_rec, goff), ("same bytes, one record per genome", d_rec1, G, np.arange(G + 1, dtype=np.uint64))):
        c = lash_amd.Context(0)
        for _ in range(3):
            c.sketch_batch_device(an, k, p, 42, d_seq, dr, nr, go, gbo, d_img)
        c.synchronize()
        c.enable_timing(True)
        for _ in range(8):
            c.sketch_batch_device(an, k, p, 42, d_seq, dr, nr, go, gbo, d_img)
        c.synchronize()
        t = c.timing()
        ms = (t["sketch_ms"] + t["finalize_ms"] + t["pack_ms"]) / 8
        print("%s k=%d %-34s %.3f ms per step -> %.4g k-mers/s, %.4g input B/s, direct launches %d of 8" %
              (an, k, name, ms, t["kmers"] / 8 / (ms * 1e-3), total / (ms * 1e-3), t["direct_launches"]), flush=True)
        c.close()
