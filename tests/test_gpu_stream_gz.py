"""BASELINE configs[4] in its stated FORM at a size the suite can afford: >= 1 Gbp of 150-bp reads as ONE
multi-member FASTQ.gz, streamed through the CLI in chunks (`lash sketch -a ull -p 12 -k 16 --stream-mb 256`: parallel member
inflate -> chunks cut at record boundaries -> device FASTQ parse -> on-device accumulation into one sketch), against
lash_sketch_batch_device on the same reads resident in HBM (a different kernel chain: records, no parse, no accumulation) and
against the oracle on three disjoint windows of reads.  The reference's counterpart: one needletail reader over the file,
utils.rs:453-459, one sketch for the whole file."""
import os
import subprocess
import sys
import time
from multiprocessing import Pool

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import host_lib as H
import oracle_lib as O

RL = 150


def test_one_gbp_fastq_gz_streamed_through_the_cli(tmp_path):
    import torch
    import lash_amd
    import make_reads_gz as M
    gbp = float(os.environ.get("LASH_TEST_STREAM_GBP", "1.0"))
    n_reads, members = int(gbp * 1e9) // RL, 32
    per = (n_reads + members - 1) // members
    jobs = [(i, min(per, n_reads - i * per)) for i in range(members) if n_reads - i * per > 0]
    path = str(tmp_path / "reads.fastq.gz")
    t0 = time.perf_counter()
    with Pool(min(16, os.cpu_count() or 1)) as pool, open(path, "wb") as f:
        for blob, _ in pool.imap(M.member, jobs):
            f.write(blob)
    t_make = time.perf_counter() - t0
    lst = tmp_path / "l.txt"
    lst.write_text(path + "\n")
    out = str(tmp_path / "reads")
    algo, k, p = "ull", 16, 12
    t0 = time.perf_counter()
    r = subprocess.run([H.CLI, "sketch", "-f", str(lst), "-o", out, "-a", algo, "-k", str(k), "-p", str(p), "--stream-mb", "256", "-t", "16"],
                       capture_output=True, text=True)
    t_cli = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    blob = H.zstd_read(out + "_sketches.bin")
    ib = lash_amd.image_bytes(algo, p)
    assert len(blob) == ib                                                # one file, one sketch

    # the same reads resident in HBM, as records of one sketch
    ctx = lash_amd.Context(0, stream=torch.cuda.current_stream())
    d_seq = torch.empty(n_reads * RL, dtype=torch.uint8, device="cuda")
    at = 0
    for idx, n in jobs:
        d_seq[at:at + n * RL].copy_(torch.from_numpy(M.member_bases(idx, n)))
        at += n * RL
    d_rec = torch.arange(0, n_reads + 1, dtype=torch.int64, device="cuda") * RL
    d_img = torch.zeros(ib, dtype=torch.uint8, device="cuda")
    goff, gbo = np.array([0, n_reads], np.uint64), np.array([0, n_reads * RL], np.uint64)
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, n_reads, goff, gbo, d_img)
    torch.cuda.synchronize()
    assert blob == d_img.cpu().numpy().tobytes(), "streamed .gz through the CLI != the same reads sketched from HBM"

    # three disjoint windows of 200 000 reads against the oracle (GPU record path == CPU restatement on exactly these reads)
    W = 200_000
    for w0 in (0, n_reads // 2 - W // 2, n_reads - W):
        host = d_seq[w0 * RL:(w0 + W) * RL].cpu().numpy()
        off = np.arange(W + 1, dtype=np.uint64) * np.uint64(RL)
        want = O.sketch_genomes(O.ULL, k, p, 42, host, off, np.array([0, W], np.uint64), threads=8)[0]
        d_w = torch.zeros(ib, dtype=torch.uint8, device="cuda")
        ctx.sketch_batch_device(algo, k, p, 42, d_seq[w0 * RL:(w0 + W) * RL], d_rec[:W + 1], W, np.array([0, W], np.uint64),
                                np.array([0, W * RL], np.uint64), d_w)
        torch.cuda.synchronize()
        assert np.array_equal(d_w.cpu().numpy(), want), "window at read %d" % w0
    ctx.close()
    print("made %.1f Gbp .gz in %.1f s, CLI %.1f s" % (gbp, t_make, t_cli))
