import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import lash_amd
for G in (4, 40, 200):
    L = 5_000_000
    ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
    d_seq = torch.empty(G * L, dtype=torch.uint8, device="cuda"); ctx.synth_genomes_device(0, G, L, d_seq)
    rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L); d_rec = torch.from_numpy(rec_off.astype(np.int64)).cuda(); goff = np.arange(G + 1, dtype=np.uint64)
    for algo, k, p in (("ull", 16, 20), ("ull", 16, 23)):
        ib = lash_amd.image_bytes(algo, p); d_img = torch.zeros(G * ib, dtype=torch.uint8, device="cuda")
        try:
            ctx.enable_timing(True)
            ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img); torch.cuda.synchronize(); tm = ctx.timing(); print(G, algo, k, p, "ok", tm["sketch_workgroups"], "%.2f ms" % tm["sketch_ms"])
        except Exception as e:
            print(G, algo, k, p, "FAIL", str(e)[-120:])
    ctx.close()
