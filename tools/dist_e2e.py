#!/usr/bin/env python3
"""tools/dist_e2e.py [n] — end-to-end `lash dist` all-vs-all on n HyperMinHash sketches of synthetic 1 Mbp genomes
(sketched here on the GPU, written in the CLI's file formats).  GPU box."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lash_amd
import host_lib as H          # liblash_host.so test hooks: zstd writer

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 1_000_000
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
ib = lash_amd.image_bytes("hmh", 0)
imgs = torch.zeros(n * ib, dtype=torch.uint8, device="cuda")
for g0 in range(0, n, 500):
    m = min(500, n - g0)
    d = torch.empty(m * L, dtype=torch.uint8, device="cuda")
    ctx.synth_genomes_device(g0, m, L, d)
    off = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
    ctx.sketch_batch_device("hmh", 16, 0, 42, d, torch.from_numpy(off.astype(np.int64)).cuda(), m, np.arange(m + 1, dtype=np.uint64), off,
                            imgs[g0 * ib:(g0 + m) * ib])
torch.cuda.synchronize()
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    H.zstd_write(os.path.join(td, "s_sketches.bin"), imgs.cpu().numpy().tobytes(), level=3, workers=8)
    json.dump(["genome_%d.fa" % i for i in range(n)], open(os.path.join(td, "s_files.json"), "w"))
    json.dump({"algorithm": "hmh", "k": "16", "molecule": "nucleotide", "seed": "42"}, open(os.path.join(td, "s_parameters.json"), "w"))
    for threads in (1, 32):
        t0 = time.perf_counter()
        r = subprocess.run([H.CLI, "dist", "-q", "s", "-r", "s", "-o", "d.txt", "-t", str(threads)], cwd=td, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        sys.stderr.write(r.stderr)
        rows = n * (n + 1) // 2
        print("lash dist -t %d: %d sketches, %d pairs in %.2f s -> %.3g pairs/s (output %.1f MB)" %
              (threads, n, rows, dt, rows / dt, os.path.getsize(os.path.join(td, "d.txt")) / 1e6))
