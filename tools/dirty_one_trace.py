#!/usr/bin/env python3
"""tools/dirty_one_trace.py B [genomes]  (the trace needs LASH_GFX950_LIB=build/variants/liblash_trace.so: tools/build_trace_lib.sh)
 — ONE soft-masked shape of tools/dirty_rate.py (every other block of B bytes lower-case; B = 0: clean), timed,
and its last launch traced per workgroup (LASH_ITEM_TRACE -> gpurun_out/item_trace_B.txt, read by tools/item_trace.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import lash_amd

B = int(sys.argv[1])
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
L, k = 5_000_000, 16
dev = torch.device("cuda:0")
ctx = lash_amd.Context(0)
d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
ctx.synth_genomes_device(0, G, L, d_seq)
ctx.synchronize()
if B:
    pos = torch.arange(L, device=dev) % (2 * B)
    d_seq.view(G, L)[:, pos >= B] |= 0x20
torch.cuda.synchronize()
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
goff = np.arange(G + 1, dtype=np.uint64)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
d_img = torch.zeros(G * lash_amd.image_bytes("hmh"), dtype=torch.uint8, device=dev)
for _ in range(3):
    ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
ctx.synchronize()
ctx.enable_timing(True)
for _ in range(10):
    ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
t = ctx.timing()
print("B=%d: direct %.3f ms  sketch %.3f ms  k-mers/step %.4g  direct launches %d  defer %d" % (B, t["direct_ms"] / 10, t["sketch_ms"] / 10, t["kmers"] / 10, t["direct_launches"], t["defer_launches"]))
out = os.path.join("gpurun_out", "item_trace_%d.txt" % B)
os.makedirs("gpurun_out", exist_ok=True)
if os.path.exists(out):
    os.remove(out)
os.environ["LASH_ITEM_TRACE"] = out
ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, G, goff, rec_off, d_img)
ctx.synchronize()
del os.environ["LASH_ITEM_TRACE"]
ctx.close()
