import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import lash_amd
G, L = int(sys.argv[1]), int(sys.argv[2])
ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda")
ctx.synth_genomes_device(0, G, L, d_seq)
rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda()
goff = np.arange(G + 1, dtype=np.uint64)
for algo, k, p in (("hmh", 16, 0), ("hll", 21, 12), ("ull", 16, 12)):
    d_img = torch.zeros(G * lash_amd.image_bytes(algo, p), dtype=torch.uint8, device="cuda")
    for _ in range(2): ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    torch.cuda.synchronize(); ctx.enable_timing(True)
    for _ in range(3): ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img)
    torch.cuda.synchronize(); tm = ctx.timing(); ctx.enable_timing(False)
    print("%s threads=%s  %d x %d: sketch stage %.2f ms" % (algo, os.environ.get("LASH_SKETCH_THREADS", "default"), G, L, tm["sketch_ms"] / 3))
