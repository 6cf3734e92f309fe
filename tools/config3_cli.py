#!/usr/bin/env python3
"""tools/config3_cli.py [N] — BASELINE configs[3]'s all-vs-all through the C++ command line at full size on one GPU box:
N (default 100 000) synthetic 5 Mbp genomes sketched in HBM (hmh k=16), written as a lash sketch-file set, then
`lash dist -q c3 -r c3 -o /dev/null` (map order and --file-order) with LASH_CLI_TIMING=1.  Prints the stage marks."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lash_amd
import host_lib as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
L, k, chunk = 5_000_000, 16, 12_500
ctx = lash_amd.Context(0)
dev = torch.device("cuda", 0)
ib = lash_amd.image_bytes("hmh")
img = torch.zeros((N, ib), dtype=torch.uint8, device=dev)
chunk = min(chunk, N)
d_seq = torch.empty(chunk * L, dtype=torch.uint8, device=dev)
t0 = time.perf_counter()
for c0 in range(0, N, chunk):
    n = min(chunk, N - c0)
    ctx.synth_genomes_device(c0, n, L, d_seq)
    rec_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
    ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, n, np.arange(n + 1, dtype=np.uint64), rec_off, img[c0:c0 + n].reshape(-1))
    ctx.synchronize()
print("sketched %d genomes in %.2f s" % (N, time.perf_counter() - t0), flush=True)
del d_seq
work = tempfile.mkdtemp(prefix="c3cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
names = ["g%06d.fa" % i for i in range(N)]
t0 = time.perf_counter()
H.zstd_write(os.path.join(work, "c3_sketches.bin"), img.cpu().numpy().tobytes(), 3, 16)
open(os.path.join(work, "c3_files.json"), "w").write(H.json_array(names))
H.write_parameters(os.path.join(work, "c3"), "hmh", k, 0, 42)
print("sketch files written in %.2f s (%.2f GB compressed)" % (time.perf_counter() - t0, os.path.getsize(os.path.join(work, "c3_sketches.bin")) / 1e9), flush=True)
ctx.close()
del img
torch.cuda.empty_cache()
for extra in ([], ["--file-order"]):
    t0 = time.perf_counter()
    r = subprocess.run([H.CLI, "dist", "-q", "c3", "-r", "c3", "-o", "/dev/null", "-t", "32"] + extra, cwd=work, capture_output=True, text=True,
                       env=dict(os.environ, LASH_CLI_TIMING="1"))
    print("lash dist %s: rc %d, %.2f s wall" % (" ".join(extra) or "(map order)", r.returncode, time.perf_counter() - t0))
    print(r.stderr[-1500:], flush=True)
for f in os.listdir(work):
    os.remove(os.path.join(work, f))
os.rmdir(work)
