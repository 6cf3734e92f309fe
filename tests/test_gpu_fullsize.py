"""Full-size GPU checks at BASELINE.json's configurations (inputs generated in HBM; the oracle cannot finish these in
seconds, so parity rests on size-independent properties of the path plus oracle spot checks on a few genomes):
  * the device k-mer census equals the reference iterator's count (L - k + 1 per genome);
  * spot-checked genomes are bit-identical to the CPU oracle;
  * idempotence: unioning a batch into its own sketches changes nothing (max / OR are idempotent);
  * permutation: sketching the genomes in another order permutes the images and nothing else;
  * splitting: a genome sketched in two halves (overlapping by k-1 bases) and merged equals the genome sketched whole;
  * streaming: cfg5's chunked accumulate path equals the one-shot sketch of the same reads.
Set LASH_FULLSIZE=0 to shrink the workloads 10x (CI boxes with little HBM)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
FULL = os.environ.get("LASH_FULLSIZE", "1") != "0"
L = 5_000_000


@pytest.fixture(scope="module")
def env():
    import torch
    import lash_amd
    ctx = lash_amd.Context(0)
    yield ctx, torch, lash_amd
    ctx.close()


def _sketch_synth(ctx, torch, lash_amd, first, G, algo, k, p, flags=0, d_img=None, d_seq=None):
    dev = torch.device("cuda", 0)
    if d_seq is None:
        d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
        ctx.synth_genomes_device(first, G, L, d_seq)
    rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    goff = np.arange(G + 1, dtype=np.uint64)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
    ib = lash_amd.image_bytes(algo, p)
    if d_img is None:
        d_img = torch.zeros(G * ib, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.enable_timing(True)
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img, flags=flags)
    t = ctx.timing()
    ctx.enable_timing(False)
    return d_seq, d_img.view(G, ib), t


@pytest.mark.parametrize("algo,k,p,G", [("hmh", 16, 0, 1000), ("hll", 21, 14, 10000)])
def test_baseline_config_properties(env, algo, k, p, G):
    """configs[1]: 1 000 x 5 Mbp hmh k=16;  configs[2]: 10 000 x 5 Mbp hll p=14 k=21 (50 GB of ASCII in HBM)."""
    ctx, torch, lash_amd = env
    if not FULL:
        G //= 10
    algo_id = lash_amd.ALGOS[algo]
    d_seq, img, t = _sketch_synth(ctx, torch, lash_amd, 0, G, algo, k, p)
    assert t["kmers"] == G * (L - k + 1) and t["bases_last"] == G * L
    if FULL and "LASH_DEFER_MIN" not in os.environ:
        # work items of 0.6 Mbp and more (here 1.67 Mbp slices): the HyperMinHash launch defers its signatures (process_word_defer), the others never do
        assert t["defer_launches"] == (1 if algo == "hmh" else 0), t
    # oracle spot checks
    for g in (0, G // 3, G - 1):
        want = O.sketch_genomes(algo_id, k, p, 42, O.synth_genome(g, L), np.array([0, L], np.uint64), np.array([0, 1], np.uint64))[0]
        assert np.array_equal(img[g].cpu().numpy(), want), g
    # distinct genomes give distinct sketches (no slot is written twice / skipped)
    assert torch.unique(img[: min(G, 512)], dim=0).shape[0] == min(G, 512)
    # idempotence
    before = img.clone()
    _, img2, _ = _sketch_synth(ctx, torch, lash_amd, 0, G, algo, k, p, flags=lash_amd.F_ACCUMULATE, d_img=img.reshape(-1), d_seq=d_seq)
    assert torch.equal(img2, before)
    del d_seq, img2
    # permutation: genomes G-1 .. 0 generated in reverse order
    n = min(G, 64)
    d_rev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    for i in range(n):
        ctx.synth_genomes_device(n - 1 - i, 1, L, d_rev[i * L:(i + 1) * L])
    ctx.synchronize()
    _, img_rev, _ = _sketch_synth(ctx, torch, lash_amd, 0, n, algo, k, p, d_seq=d_rev)
    assert torch.equal(img_rev.flip(0), before[:n])


@pytest.mark.parametrize("algo,k,p", [("hmh", 16, 0), ("hll", 21, 14), ("ull", 16, 12)])
def test_split_and_merge_equals_whole(env, algo, k, p):
    ctx, torch, lash_amd = env
    dev = torch.device("cuda", 0)
    G = 8
    d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(500, G, L, d_seq)
    _, whole, _ = _sketch_synth(ctx, torch, lash_amd, 500, G, algo, k, p, d_seq=d_seq)
    # two "genomes" per genome: [0, cut + k - 1) and [cut, L): together they contain every k-mer exactly as the whole does
    cut = 2_345_679
    halves = torch.cat([torch.cat([d_seq[g * L:g * L + cut + k - 1], d_seq[g * L + cut:(g + 1) * L]]) for g in range(G)])
    lens = []
    for g in range(G):
        lens += [cut + k - 1, L - cut]
    rec_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    goff = np.arange(2 * G + 1, dtype=np.uint64)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
    ib = lash_amd.image_bytes(algo, p)
    d_img = torch.zeros(2 * G * ib, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.sketch_batch_device(algo, k, p, 42, halves, d_rec, 2 * G, goff, rec_off, d_img)
    ctx.synchronize()
    parts = d_img.view(G, 2, ib)
    a = parts[:, 0].contiguous()
    b = parts[:, 1].contiguous()
    ctx.merge_images_device(algo, p, a, b, G)
    ctx.synchronize()
    assert torch.equal(a, whole)


def test_streamed_metagenome_chunks(env):
    """configs[4] shape: 150-bp reads streamed in chunks into ONE ull p=12 sketch with on-device accumulation.
    Full size would be 100 Gbp; the property (chunked == one-shot) is checked at 3 Gbp (0.3 Gbp when shrunk)."""
    ctx, torch, lash_amd = env
    dev = torch.device("cuda", 0)
    total = 3_000_000_000 if FULL else 300_000_000
    total -= total % 150
    n_reads = total // 150
    d_seq = torch.empty(total, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(900_000, 1, total, d_seq)          # one long random sequence cut into 150-bp reads
    ctx.synchronize()
    ib = lash_amd.image_bytes("ull", 12)

    def run(chunks):
        d_img = torch.zeros(ib, dtype=torch.uint8, device=dev)
        per = (n_reads + chunks - 1) // chunks
        kmers = 0
        for c in range(chunks):
            r0, r1 = c * per, min(n_reads, (c + 1) * per)
            if r1 <= r0:
                break
            d_rec = (torch.arange(r0, r1 + 1, dtype=torch.int64, device=dev) * 150)
            goff = np.array([0, r1 - r0], dtype=np.uint64)
            gbo = np.array([r0 * 150, r1 * 150], dtype=np.uint64)
            torch.cuda.synchronize()
            ctx.enable_timing(True)
            # record offsets are absolute into d_seq; genome_rec_off indexes d_rec
            ctx.sketch_batch_device("ull", 16, 12, 42, d_seq, d_rec, r1 - r0, goff, gbo, d_img,
                                    flags=lash_amd.F_ACCUMULATE if c else 0)
            kmers += ctx.timing()["kmers"]
            ctx.enable_timing(False)
        return d_img, kmers

    one, k1 = run(1)
    many, k2 = run(7)
    assert k1 == k2 == n_reads * (150 - 16 + 1)
    assert torch.equal(one, many)
    # oracle on a prefix of the reads, merged with itself, is a sub-sketch: every register it sets is <= in the full sketch's union
    head = 200_000
    host = d_seq[: head * 150].cpu().numpy()
    off = (np.arange(head + 1, dtype=np.uint64) * 150)
    sub = O.sketch_genomes(O.ULL, 16, 12, 42, host, off, np.array([0, head], np.uint64), threads=8)[0]
    merged = O.merge_images(O.ULL, 12, one.cpu().numpy(), sub)
    assert np.array_equal(merged, one.cpu().numpy())


def test_config4_full_100gbp(env):
    """BASELINE configs[4] at its stated size: ull p=12 k=16 over 100 Gbp of 150-bp reads (6.67e8 records), ASCII resident
    in HBM (100 GB), streamed through lash_sketch_batch_device in chunks of <= 2^32-64 bytes with on-device accumulation
    (utils.rs:453-459: one sketch per file, however many records).  The oracle cannot finish 9e10 k-mers, so:
      * census: the device counts exactly n_reads * (150 - 16 + 1) k-mers;
      * two different chunkings give the identical image;
      * three disjoint windows of 1 M reads are sketched alone on the GPU and must equal the oracle bit for bit, and the
        oracle's window sketches merged into the full sketch change nothing (every register they set is in it)."""
    ctx, torch, lash_amd = env
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    total = 100_000_000_050 if FULL else 3_000_000_000
    if free < total + 24 * 2**30:
        pytest.skip("needs %.0f GB of free HBM" % ((total + 24 * 2**30) / 1e9))
    RL, k, p = 150, 16, 12
    total -= total % RL
    n_reads = total // RL
    d_seq = torch.empty(total, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(910_000, 1, total, d_seq)
    ctx.synchronize()
    ib = lash_amd.image_bytes("ull", p)

    def run(chunks):
        d_img = torch.zeros(ib, dtype=torch.uint8, device=dev)
        per = (n_reads + chunks - 1) // chunks
        assert per * RL <= 0xFFFFFFFF - 64
        torch.cuda.synchronize()
        ctx.enable_timing(True)
        for c in range(chunks):
            r0, r1 = c * per, min(n_reads, (c + 1) * per)
            if r1 <= r0:
                break
            d_rec = torch.arange(r0, r1 + 1, dtype=torch.int64, device=dev) * RL
            torch.cuda.synchronize()
            ctx.sketch_batch_device("ull", k, p, 42, d_seq, d_rec, r1 - r0, np.array([0, r1 - r0], np.uint64),
                                    np.array([r0 * RL, r1 * RL], np.uint64), d_img, flags=lash_amd.F_ACCUMULATE if c else 0)
            ctx.synchronize()
        t = ctx.timing()
        ctx.enable_timing(False)
        return d_img, t

    a, ta = run(25 if FULL else 3)
    b, tb = run(37 if FULL else 5)
    assert ta["kmers"] == tb["kmers"] == n_reads * (RL - k + 1)
    assert torch.equal(a, b)
    full = a.cpu().numpy()
    assert int(np.frombuffer(full[:8].tobytes(), "<u8")[0]) == 1 << p and (full[8:] != 0).all()   # 9e10 k-mers fill every register
    win = 1_000_000
    for w0 in (0, n_reads // 2 - 333_333, n_reads - win):
        d_rec = torch.arange(w0, w0 + win + 1, dtype=torch.int64, device=dev) * RL
        d_one = torch.zeros(ib, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.sketch_batch_device("ull", k, p, 42, d_seq, d_rec, win, np.array([0, win], np.uint64),
                                np.array([w0 * RL, (w0 + win) * RL], np.uint64), d_one)
        ctx.synchronize()
        host = d_seq[w0 * RL:(w0 + win) * RL].cpu().numpy()
        want = O.sketch_genomes(O.ULL, k, p, 42, host, np.arange(win + 1, dtype=np.uint64) * RL, np.array([0, win], np.uint64))[0]
        assert np.array_equal(d_one.cpu().numpy(), want), w0
        assert np.array_equal(O.merge_images(O.ULL, p, full, want), full), w0
    print("configs[4] full size: %d reads, %.3e k-mers; device time pack %.1f + sketch %.1f + finalize %.1f ms over %d calls"
          % (n_reads, ta["kmers"], ta["pack_ms"], ta["sketch_ms"], ta["finalize_ms"], ta["calls"]))


def test_config3_full_100k_genomes_all_vs_all(env, tmp_path):
    """BASELINE configs[3] at its stated TOTAL size on ONE GPU — 100 000 x 5 Mbp, hmh k=16, then the 100 000 x 100 000 all-vs-all
    (5.00005e9 printed pairs) through lash_amd.allpairs at world_size 1 (the 8-GPU run shards exactly this 8 ways and adds the
    all-gather; RCCL with >= 2 ranks is NOT exercised here).  Sketching runs in 8 chunks of 12 500 genomes (62.5 GB of ASCII
    resident per chunk), the images accumulate to 3.28 GB; the text goes to os.devnull.  Checked:
      * census: every chunk counts genomes x (L - k + 1) k-mers; three genomes equal the oracle bit for bit;
      * the stream prints exactly N (N + 1) / 2 rows (byte count of the fixed-width rows);
      * pair statistics: a block's diagonal has C = N = 16 384 (a full sketch against itself), a sampled off-diagonal block is
        symmetric (rows A x columns B == (rows B x columns A) transposed) and equals numpy on its first entries;
      * a 2 000-genome sub-collection through lash_amd.allpairs (real files) == `lash dist --file-order` on the same sketches, byte
        for byte (LASH_FULLSIZE=0 shrinks N to 8 000 and the sub-collection to 500)."""
    import subprocess
    import time
    import torch.distributed as dist
    import host_lib as H
    from lash_amd.allpairs import all_vs_all_stream
    ctx, torch, lash_amd = env
    dev = torch.device("cuda", 0)
    N, chunk, sub = (100_000, 12_500, 2_000) if FULL else (8_000, 1_000, 500)
    free, _ = torch.cuda.mem_get_info()
    if free < chunk * L + N * 40_000 * 4 + 24 * 2**30:
        pytest.skip("needs %.0f GB of free HBM" % ((chunk * L + N * 40_000 * 4 + 24 * 2**30) / 1e9))
    k, ib = 16, lash_amd.image_bytes("hmh")
    img = torch.zeros((N, ib), dtype=torch.uint8, device=dev)
    d_seq = torch.empty(chunk * L, dtype=torch.uint8, device=dev)
    rec_off = np.arange(chunk + 1, dtype=np.uint64) * np.uint64(L)
    goff = np.arange(chunk + 1, dtype=np.uint64)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
    t0 = time.perf_counter()
    sketch_ms = 0.0
    for c in range(N // chunk):
        ctx.synth_genomes_device(c * chunk, chunk, L, d_seq)
        ctx.synchronize()
        torch.cuda.synchronize()
        ctx.enable_timing(True)
        ctx.sketch_batch_device("hmh", k, 0, 42, d_seq, d_rec, chunk, goff, rec_off, img[c * chunk:(c + 1) * chunk].reshape(-1))
        t = ctx.timing()
        ctx.enable_timing(False)
        assert t["kmers"] == chunk * (L - k + 1) and t["bases_last"] == chunk * L, c
        sketch_ms += t["pack_ms"] + t["sketch_ms"] + t["finalize_ms"]
    t_sketch = time.perf_counter() - t0
    for g in (0, N // 2 + 7, N - 1):
        want = O.sketch_genomes(O.HMH, k, 0, 42, O.synth_genome(g, L), np.array([0, L], np.uint64), np.array([0, 1], np.uint64))[0]
        assert np.array_equal(img[g].cpu().numpy(), want), g
    del d_seq
    torch.cuda.empty_cache()

    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        names = ["g%06d.fa" % i for i in range(N)]
        stats = {}
        t0 = time.perf_counter()
        parts = all_vs_all_stream("hmh", 0, k, img, [N], names, os.devnull, ctx=ctx, stats=stats)
        t_all = time.perf_counter() - t0
        line = 2 * len(names[0]) + 2 + 8 + 1                                     # "ref\tqry\t0.123456\n"
        assert sum(b for _, _, b in parts) == N * (N + 1) // 2 * line
        assert stats["printed_pairs"] == N * (N + 1) // 2
        # the small sub-collection with real files against the C++ command line on the same sketches
        idx = torch.arange(0, N, N // sub, device=dev)[:sub]
        sub_img = img[idx].contiguous()
        sub_names = [names[int(i)] for i in idx.cpu()]
        out = str(tmp_path / "sub.tsv")
        bands = all_vs_all_stream("hmh", 0, k, sub_img, [sub], sub_names, out, ctx=ctx)
        body = b"".join(open(pth, "rb").read() for _, pth, _ in sorted(bands))
        H.zstd_write(str(tmp_path / "sub_sketches.bin"), sub_img.cpu().numpy().tobytes())
        (tmp_path / "sub_files.json").write_text(H.json_array(sub_names))
        H.write_parameters(str(tmp_path / "sub"), "hmh", k, 0, 42)
        r = subprocess.run([H.CLI, "dist", "-q", "sub", "-r", "sub", "-o", "cli.tsv", "--file-order", "-t", "8"], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert (tmp_path / "cli.tsv").read_bytes() == b"Reference\tQuery\tDistance\n" + body
    finally:
        if own_pg:
            dist.destroy_process_group()
    # pair statistics at full size, straight from the resident set
    s = ctx.sketch_set("hmh", 0, img)
    s.prepare()
    r0 = N - 777
    tri = s.pair_block(r0, r0 + 300, n_cols=r0 + 300, triangle=True)
    for i in range(300):
        assert tri["c_or_zero"][i, r0 + i] == 16384 and tri["n_counts"][i, r0 + i] == 16384
    a0, b0 = N // 2 + 1000, 123
    ab = s.pair_block(a0, a0 + 200, n_cols=b0 + 600)["c_or_zero"][:, b0:b0 + 600]
    ba = s.pair_block(b0, b0 + 600, n_cols=a0 + 200)["c_or_zero"][:, a0:a0 + 200]
    assert np.array_equal(ab, ba.T)
    x = img[a0:a0 + 2].cpu().numpy().view(np.uint16)
    y = img[b0:b0 + 3].cpu().numpy().view(np.uint16)
    for i in range(2):
        for j in range(3):
            assert ab[i, j] == int(((x[i] == y[j]) & (x[i] != 0)).sum())
    s.free()
    print("configs[3] total size on one GPU: %d genomes sketched in %.2f s wall (%.1f ms of device stages); all-vs-all %.1f s: gather %.2f, "
          "cardinalities + operands %.2f, pair blocks incl. copy back %.2f, host rows (lash_dist_rows + text, %d threads) %.2f; %.3e printed pairs"
          % (N, t_sketch, sketch_ms, t_all, stats["gather_s"], stats["prepare_s"], stats["pair_blocks_s"], stats["threads"], stats["host_rows_s"],
             stats["printed_pairs"]))


@pytest.mark.sole
@pytest.mark.parametrize("algo,k,p", [("hmh", 16, 0), ("hll", 21, 10)])
def test_a_viral_collection_at_full_size(env, algo, k, p, monkeypatch):
    """Round 5's shape at the size it was measured on (tools/viral_rate.py): 200 000 genomes of 3..300 kbp (log-uniform), 1..4 records each,
    12.9 GB of ASCII resident — whole through the persistent small-genome kernel (sole_kernels.hip).  Size-independent checks: the k-mer
    census equals the sum over records of max(0, len - k + 1); the SLICED kernels (LASH_F_NO_SOLE: round 4's route, a different kernel,
    different planning) give the same bytes for every genome; 240 genomes spread over the collection — among them the smallest, the
    largest, the first and the last — equal the oracle bit for bit."""
    ctx, torch, lash_amd = env
    monkeypatch.setenv("LASH_SOLE_MAX", "393216")
    G = 200_000 if FULL else 20_000
    rng = np.random.default_rng(13)
    lens = np.exp(rng.uniform(np.log(3e3), np.log(3e5), size=G)).astype(np.int64)
    gbo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(gbo[-1])
    nrec = rng.integers(1, 5, size=G)
    dev = torch.device("cuda", 0)
    d_seq = torch.empty(total, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(0, 1, total, d_seq)                      # one long synthetic sequence, cut into the genomes
    # record starts: sorted cut points inside every genome
    starts = [gbo[:-1].astype(np.int64)]
    cuts = (rng.random(size=(G, 3)) * (lens[:, None] - 1)).astype(np.int64) + 1 + gbo[:-1, None].astype(np.int64)
    keep = np.arange(3)[None, :] < (nrec[:, None] - 1)
    rec_off = np.unique(np.concatenate([starts[0], cuts[keep], [total]])).astype(np.uint64)
    goff = np.searchsorted(rec_off, gbo).astype(np.uint64)             # first record of every genome (its byte offset IS a record start)
    n_rec = len(rec_off) - 1
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
    ib = lash_amd.image_bytes(algo, p)
    d_img = torch.zeros(G * ib, dtype=torch.uint8, device=dev)
    ctx.enable_timing(True)
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, n_rec, goff, gbo, d_img)
    ctx.synchronize()
    t = ctx.timing()
    ctx.enable_timing(False)
    rl = np.diff(rec_off.astype(np.int64))
    assert t["sole_launches"] == 1 and t["kmers"] == int(np.maximum(rl - k + 1, 0).sum()) and t["bases_last"] == total
    # the other route, every genome
    d_img2 = torch.zeros(G * ib, dtype=torch.uint8, device=dev)
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, n_rec, goff, gbo, d_img2, flags=lash_amd.F_NO_SOLE)
    ctx.synchronize()
    assert torch.equal(d_img, d_img2)
    del d_img2
    # the oracle on a spread of genomes
    pick = sorted(set([0, G - 1, int(np.argmin(lens)), int(np.argmax(lens))] + [int(x) for x in rng.integers(0, G, size=236)]))
    img = d_img.view(G, ib)
    algo_id = lash_amd.ALGOS[algo]
    for g in pick:
        host = d_seq[int(gbo[g]):int(gbo[g + 1])].cpu().numpy()
        ro = rec_off[int(goff[g]):int(goff[g + 1]) + 1] - gbo[g]
        want = O.sketch_genomes(algo_id, k, p, 42, host, ro.astype(np.uint64), np.array([0, len(ro) - 1], np.uint64))[0]
        assert np.array_equal(img[g].cpu().numpy(), want), (algo, g, int(lens[g]))


@pytest.mark.sole
@pytest.mark.parametrize("algo,k,p", [("hll", 21, 10), ("hmh", 16, 0), ("ull", 16, 10)])
def test_a_million_tiny_genomes(env, algo, k, p, monkeypatch):
    """10^6 genomes of 150..2 500 bytes (amplicons, short contigs), every fourth one with a run of N and lower case, 1.3 GB resident: the
    persistent kernel with 250+ genomes per workgroup.  Census, the sliced route on every genome, 300 genomes against the oracle."""
    ctx, torch, lash_amd = env
    monkeypatch.setenv("LASH_SOLE_MAX", "393216")
    G = 1_000_000 if FULL else 100_000
    rng = np.random.default_rng(17)
    lens = rng.integers(150, 2501, size=G).astype(np.int64)
    gbo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(gbo[-1])
    dev = torch.device("cuda", 0)
    d_seq = torch.empty(total, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(0, 1, total, d_seq)
    ctx.synchronize()                    # (the generator runs on the context's stream, the edits below on torch's: without this some of them were
                                         #  overwritten by the generator once in a while — bases_last 0.02 % high, round 6)
    # deleted bytes: in every fourth genome bytes [40, 40 + 12) become N and [90, 90 + 25) lower case
    dirty = torch.from_numpy(gbo[:-1:4].astype(np.int64)).to(dev)
    for o, n, orv in ((40, 12, None), (90, 25, 0x20)):
        idx = (dirty[:, None] + torch.arange(o, o + n, device=dev)[None, :]).reshape(-1)
        if orv is None:
            d_seq[idx] = 0x4E
        else:
            d_seq[idx] |= orv
    d_rec = torch.from_numpy(gbo.astype(np.int64)).to(dev)
    goff = np.arange(G + 1, dtype=np.uint64)
    ib = lash_amd.image_bytes(algo, p)
    d_img = torch.zeros(G * ib, dtype=torch.uint8, device=dev)
    ctx.enable_timing(True)
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, gbo, d_img)
    ctx.synchronize()
    t = ctx.timing()
    ctx.enable_timing(False)
    n_dirty = len(gbo[:-1:4])
    surv = lens.copy()
    surv[::4] -= 37
    assert t["sole_launches"] == 1 and t["bases_last"] == total - 37 * n_dirty and t["kmers"] == int(np.maximum(surv - k + 1, 0).sum())
    d_img2 = torch.zeros(G * ib, dtype=torch.uint8, device=dev)
    ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, gbo, d_img2, flags=lash_amd.F_NO_SOLE)
    ctx.synchronize()
    assert torch.equal(d_img, d_img2)
    del d_img2
    img = d_img.view(G, ib)
    algo_id = lash_amd.ALGOS[algo]
    for g in sorted(set([0, 1, 2, 3, 4, G - 1, G - 2] + [int(x) for x in rng.integers(0, G, size=293)])):
        host = d_seq[int(gbo[g]):int(gbo[g + 1])].cpu().numpy()
        want = O.sketch_genomes(algo_id, k, p, 42, host, np.array([0, len(host)], np.uint64), np.array([0, 1], np.uint64))[0]
        assert np.array_equal(img[g].cpu().numpy(), want), (algo, g, int(lens[g]))


@pytest.mark.parametrize("algo,k,p", [("hmh", 16, 0), ("hll", 21, 14), ("ull", 16, 12)])
def test_soft_masked_assemblies_at_full_size(env, algo, k, p, monkeypatch):
    """300 genomes of 5 Mbp under a RepeatMasker-like mask — alternating upper / lower-case runs of 30..30 000 bytes at arbitrary byte
    positions, a different phase per genome, plus a run of N per genome — 1.5 GB resident.  Three routes must give the same bytes for
    every genome: the optimistic direct pass with its hand-over to stream_sketch_kernel (what a fresh context does), every genome straight
    through stream_sketch_kernel (LASH_F_STREAM_ONLY: what a context does after a few soft-masked batches; HyperMinHash: the deferring
    variant), and the pack stage first (LASH_F_NO_DIRECT); the census and the surviving-base count are exact; three genomes equal the oracle."""
    ctx, torch, lash_amd = env
    G = 300 if FULL else 30
    dev = torch.device("cuda", 0)
    d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
    ctx.synth_genomes_device(0, G, L, d_seq)
    ctx.synchronize()                    # (the edits below run on torch's stream)
    rng = np.random.default_rng(23)
    edges = np.cumsum(np.exp(rng.uniform(np.log(30), np.log(30000), size=4 * (2 * L) // 3000)).astype(np.int64))
    edges = edges[edges < 2 * L]
    lower = np.zeros(2 * L + 1, np.int8)
    lower[edges[0::2]] += 1
    lower[edges[1::2]] -= 1
    mask2 = torch.from_numpy(np.cumsum(lower[:2 * L]).astype(np.bool_)).to(dev)          # twice a genome long: every genome takes its own window
    v = d_seq.view(G, L)
    phase = rng.integers(0, L, size=G)
    for g in range(G):
        v[g, mask2[int(phase[g]):int(phase[g]) + L]] |= 0x20
        a = int(rng.integers(0, L - 5000))
        v[g, a:a + int(rng.integers(1, 4000))] = 0x4E
    surviving = ((v == 0x41) | (v == 0x43) | (v == 0x47) | (v == 0x54)).sum(dim=1).cpu().numpy().astype(np.int64)
    rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
    goff = np.arange(G + 1, dtype=np.uint64)
    d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
    ib = lash_amd.image_bytes(algo, p)
    imgs = {}
    for name, flags in (("direct + hand-over", 0), ("stream only", lash_amd.F_STREAM_ONLY), ("pack first", lash_amd.F_NO_DIRECT)):
        d_img = torch.zeros(G * ib, dtype=torch.uint8, device=dev)
        ctx.enable_timing(True)
        ctx.sketch_batch_device(algo, k, p, 42, d_seq, d_rec, G, goff, rec_off, d_img, flags=flags)
        ctx.synchronize()
        t = ctx.timing()
        ctx.enable_timing(False)
        assert t["kmers"] == int(np.maximum(surviving - k + 1, 0).sum()) and t["bases_last"] == int(surviving.sum()), (name, t)
        imgs[name] = d_img
    assert torch.equal(imgs["direct + hand-over"], imgs["pack first"]) and torch.equal(imgs["stream only"], imgs["pack first"])
    img = imgs["stream only"].view(G, ib)
    algo_id = lash_amd.ALGOS[algo]
    for g in (0, G // 2, G - 1):
        host = v[g].cpu().numpy()
        want = O.sketch_genomes(algo_id, k, p, 42, host, np.array([0, L], np.uint64), np.array([0, 1], np.uint64))[0]
        assert np.array_equal(img[g].cpu().numpy(), want), (algo, g)
