#!/usr/bin/env python3
"""tools/debug_viral.py G — the viral-collection shape at G genomes through both routes (persistent kernel / sliced kernels): census and the first 200
genomes against the oracle; for a genome that differs, which record starts the images behave as if they had lost (how round 5's stale ring pointer was
narrowed down: DESIGN 4.6 "Two bugs").  GPU box."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch, lash_amd
import oracle_lib as O
G = int(sys.argv[1]); k = 16
rng = np.random.default_rng(13)
lens = np.exp(rng.uniform(np.log(3e3), np.log(3e5), size=G)).astype(np.int64)
gbo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
total = int(gbo[-1])
nrec = rng.integers(1, 5, size=G)
ctx = lash_amd.Context(0)
d_seq = torch.empty(total, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, 1, total, d_seq)
cuts = (rng.random(size=(G, 3)) * (lens[:, None] - 1)).astype(np.int64) + 1 + gbo[:-1, None].astype(np.int64)
keep = np.arange(3)[None, :] < (nrec[:, None] - 1)
rec_off = np.unique(np.concatenate([gbo[:-1].astype(np.int64), cuts[keep], [total]])).astype(np.uint64)
goff = np.searchsorted(rec_off, gbo).astype(np.uint64)
n_rec = len(rec_off) - 1
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
host = d_seq.cpu().numpy()
print("non-ACGT bytes:", int((~np.isin(host, np.frombuffer(b"ACGT", np.uint8))).sum()), "total", total, "records", n_rec)
rl = np.diff(rec_off.astype(np.int64))
want_k = int(np.maximum(rl - k + 1, 0).sum())
for flags, name in ((0, "sole"), (lash_amd.F_NO_SOLE, "sliced")):
    d_img = torch.zeros(G * 32768, dtype=torch.uint8, device="cuda")
    ctx.enable_timing(True)
    ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, n_rec, goff, gbo, d_img, flags=flags)
    ctx.synchronize()
    t = ctx.timing(); ctx.enable_timing(False)
    print(name, "kmers", t["kmers"], "want", want_k, "diff", t["kmers"] - want_k, "bases", t["bases_last"], "sole_launches", t["sole_launches"])
    img = d_img.view(G, 32768).cpu().numpy()
    bad = 0
    for g in range(min(G, 200)):
        ro = rec_off[int(goff[g]):int(goff[g + 1]) + 1] - gbo[g]
        want = O.sketch_genomes(O.HMH, k, 0, 42, host[int(gbo[g]):int(gbo[g + 1])], ro.astype(np.uint64), np.array([0, len(ro) - 1], np.uint64))[0]
        if not np.array_equal(img[g], want):
            bad += 1
            if bad <= 12 and name == "sole":
                # which record boundaries were lost?  try dropping each subset of the inner boundaries
                inner = list(ro[1:-1])
                hit = None
                import itertools
                for r in range(1, len(inner) + 1):
                    for drop in itertools.combinations(range(len(inner)), r):
                        ro2 = np.array([ro[0]] + [x for j, x in enumerate(inner) if j not in drop] + [ro[-1]], np.uint64)
                        w2 = O.sketch_genomes(O.HMH, k, 0, 42, host[int(gbo[g]):int(gbo[g + 1])], ro2, np.array([0, len(ro2) - 1], np.uint64))[0]
                        if np.array_equal(img[g], w2): hit = [int(inner[j]) for j in drop]
                print("genome", g, "start", int(gbo[g]), "len", int(lens[g]), "record starts", [int(x) for x in ro[:-1]], "-> equals the oracle without the starts at", hit,
                      [(int(x) % 8192, (int(gbo[g]) + int(x)) % 32) for x in (hit or [])])
    print(name, "genomes differing from the oracle among the first 200:", bad)
