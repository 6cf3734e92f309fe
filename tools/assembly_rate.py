#!/usr/bin/env python3
"""tools/assembly_rate.py [total_bp=3000000000] — ONE chromosome-scale assembly as `lash sketch` meets it: 24 records
("chromosomes", 50 .. 250 Mbp), N gaps of 1 kb .. 5 Mb (log-uniform, ~4 % of the bytes: centromeres, scaffolding gaps) and RepeatMasker-like
soft masking (alternating upper / lower-case runs of 30 .. 30 000 bytes, ~50 % lower case).  filter_out_n deletes both (utils.rs:33-41);
one sketch for the whole file (utils.rs:450-509).  Prints the surviving k-mers per second next to the same bytes clean, for hmh k=16, hll p=14 k=21
and ull p=12 k=16, checks the census against the mask, and that the default route and LASH_F_STREAM_ONLY give the same image."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import lash_amd

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
dev = torch.device("cuda:0")
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
rng = np.random.default_rng(20261005)
# records
w = rng.uniform(50, 250, size=24)
rec_len = np.maximum((w / w.sum() * TOTAL).astype(np.int64), 1000)
rec_len[-1] += TOTAL - rec_len.sum()
rec_off = np.concatenate([[0], np.cumsum(rec_len)]).astype(np.uint64)
d_seq = torch.empty(TOTAL, dtype=torch.uint8, device=dev)
ctx.synth_genomes_device(4242, 1, TOTAL, d_seq)
ctx.synchronize()
d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
goff = np.array([0, len(rec_len)], np.uint64)
gbo = np.array([0, TOTAL], np.uint64)


def runs(lo, hi, mean_gap_factor, target_frac):
    """alternating keep / mark runs: mark runs log-uniform in [lo, hi], keep runs scaled so that about target_frac of the bytes are marked"""
    n = int(TOTAL / (np.exp((np.log(lo) + np.log(hi)) / 2)) * 2) + 16
    mark = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n))
    keep = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n)) * mean_gap_factor
    scale = (mark.mean() * (1 - target_frac)) / (keep.mean() * target_frac)
    edges = np.cumsum(np.stack([keep * scale, mark], axis=1).reshape(-1)).astype(np.int64)
    return edges[edges < TOTAL]


def apply(edges, fn):
    """fn(mask) on the byte ranges [edges[1], edges[2]), [edges[3], edges[4]), ...; in pieces of 256 MB"""
    marked = 0
    for a in range(0, TOTAL, 1 << 28):
        b = min(TOTAL, a + (1 << 28))
        tog = np.zeros(b - a + 1, np.int8)
        lo = np.searchsorted(edges, a, "left")
        state = lo & 1                                    # an odd number of edges before a: inside a marked run
        e = edges[lo:np.searchsorted(edges, b, "left")] - a
        np.add.at(tog, e, 1)
        m = torch.from_numpy(((np.cumsum(tog[:-1]) + state) & 1).astype(np.bool_)).to(dev)
        marked += int(m.sum())
        fn(d_seq[a:b], m)
        del m
    return marked


clean = None
soft = apply(runs(30, 30_000, 1.0, 0.5), lambda v, m: v.__ior__(m.to(torch.uint8) * 0x20))
gap_edges = runs(1_000, 5_000_000, 1.0, 0.04)
gaps = apply(gap_edges, lambda v, m: v.masked_fill_(m, ord("N")))
torch.cuda.synchronize()
# surviving bases per record -> k-mers (the oracle's count: windows of the filtered record)
keep_mask_counts = []
for r in range(len(rec_len)):
    a, b = int(rec_off[r]), int(rec_off[r + 1])
    v = d_seq[a:b]
    keep_mask_counts.append(int(((v == 65) | (v == 67) | (v == 71) | (v == 84)).sum()))
surv = np.array(keep_mask_counts, np.int64)
print("assembly: %d bp in %d records; %.1f %% lower case, %.1f %% N (gaps of 1 kb .. 5 Mb: %d of them); %.1f %% of the bytes survive filter_out_n"
      % (TOTAL, len(rec_len), 100.0 * soft / TOTAL, 100.0 * gaps / TOTAL, len(gap_edges) // 2, 100.0 * surv.sum() / TOTAL), flush=True)

for algo, k, p in (("hmh", 16, 0), ("hll", 21, 14), ("ull", 16, 12)):
    want = int(np.maximum(surv - k + 1, 0).sum())
    ib = lash_amd.image_bytes(algo, p)
    d_a = torch.zeros(ib, dtype=torch.uint8, device=dev)
    d_b = torch.zeros(ib, dtype=torch.uint8, device=dev)
    for flags, name, d_img in ((0, "default route", d_a), (lash_amd.F_STREAM_ONLY, "stream kernel only", d_b)):
        for _ in range(2):
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, len(rec_len), goff, gbo, d_img, flags=flags)
        ctx.synchronize()
        ctx.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, len(rec_len), goff, gbo, d_img, flags=flags)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 4
        tm = ctx.timing()
        ctx.enable_timing(False)
        assert tm["kmers"] == 4 * want, (tm["kmers"] // 4, want)
        print("  %s k=%d%s  %-18s %8.3f ms  %.4g surviving k-mers/s  %.4g bytes/s  (sketch stage %.3f ms; direct launches %d, census = the mask's)"
              % (algo, k, "" if algo == "hmh" else " p=%d" % p, name, dt * 1e3, want / dt, TOTAL / dt, tm["sketch_ms"] / 4, tm["direct_launches"]), flush=True)
    assert torch.equal(d_a, d_b), "the two routes disagree"
print("routes agree on every image")
