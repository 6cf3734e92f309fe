"""world_size-2 (and 3) gloo tests of the multi-GPU host logic: contiguous byte-balanced sharding of the file list and
the rank-order gather of sketch images.  The GPU path uses the same functions over RCCL ("nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lash_amd.shard import gather_images, merge_partial_images, shard_genomes


def test_shard_genomes_properties():
    rng = np.random.default_rng(1)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 2, 7, 100, 1000):
            lens = rng.integers(0, 10_000_000, size=n)
            blocks = shard_genomes(lens, world)
            assert len(blocks) == world and blocks[0][0] == 0 and blocks[-1][1] == n
            for (a, b), (c, d) in zip(blocks[:-1], blocks[1:]):
                assert a <= b == c <= d
            if n >= 50 * world:
                tot = [int(lens[a:b].sum()) for a, b in blocks]
                assert max(tot) - min(tot) <= 2 * int(lens.max()), (world, n, tot)
    # equal genomes -> equal blocks (cfg4: 100 000 genomes over 8 GPUs = 12 500 each)
    assert shard_genomes([5_000_000] * 100_000, 8) == [(i * 12_500, (i + 1) * 12_500) for i in range(8)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_image(genome_id, ib):
    # deterministic stand-in for a sketch image: depends only on the genome id
    return (np.arange(ib, dtype=np.uint64) * 2654435761 + genome_id * 40503).astype(np.uint8)


def _worker(rank, world, port, lens, ib, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blocks = shard_genomes(lens, world)
        a, b = blocks[rank]
        local = torch.from_numpy(np.stack([_fake_image(g, ib) for g in range(a, b)]) if b > a else np.zeros((0, ib), np.uint8))
        counts = [e - s for s, e in blocks]
        allimg = gather_images(local, counts)
        want = np.stack([_fake_image(g, ib) for g in range(len(lens))])
        ok = allimg.shape == (len(lens), ib) and np.array_equal(allimg.numpy(), want)
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lens", [(2, [5, 1, 1, 1, 9, 2, 3]), (2, [10] * 9), (3, [7, 7, 0, 0, 100, 3, 3, 3]), (2, [4])])
def test_gather_images_in_file_order_gloo(world, lens):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lens, 257, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert all(res[r] for r in range(world)), res


def _hmh_union_cpu(dst, src):
    # test stand-in for lash_merge_images on HyperMinHash images: register-wise max of the u16 registers
    a = dst.numpy().view(np.uint16)
    np.maximum(a, src.numpy().view(np.uint16), out=a)


def _merge_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        local = torch.from_numpy(rng.integers(0, 60000, size=(3, 64), dtype=np.uint16).view(np.uint8).copy())
        merged = merge_partial_images(local, _hmh_union_cpu)
        want = np.zeros((3, 64), np.uint16)
        for r in range(world):
            want = np.maximum(want, np.random.default_rng(100 + r).integers(0, 60000, size=(3, 64), dtype=np.uint16))
        q.put((rank, bool(np.array_equal(merged.numpy().view(np.uint16), want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_merge_partial_images_gloo(world):
    """configs[4] shape: every rank sketched a different chunk of the same inputs; all ranks end with the union."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert all(res[r] for r in range(world)), res


# ---- all-vs-all dist over ranks (BASELINE configs[3]; lash_amd/allpairs.py): REAL sketch images (made by the oracle: no GPU
# here), gathered over gloo, every rank computes its block of reference rows; the union of the blocks must be the
# single-process matrix.  The pair statistics — GPU kernels in the product — are a numpy stand-in here (test infrastructure);
# everything else (shard, gather, row blocks, cardinalities, lash_dist_rows) is the product's code.
def _numpy_pair_stats(algo, p, estimator, ref, qry):
    import lash_amd
    import oracle_lib as O
    if algo == "hmh":
        a, b = ref.view("<u2").astype(np.int64), qry.view("<u2").astype(np.int64)
        c = ((a[:, None, :] == b[None, :, :]) & (a[:, None, :] != 0)).sum(axis=2).astype(np.uint32)
        n = ((a[:, None, :] != 0) | (b[None, :, :] != 0)).sum(axis=2).astype(np.uint32)
        return dict(c_or_zero=c, n_counts=n)
    if algo == "hll":
        u = np.maximum(ref[:, None, 33:], qry[None, :, 33:]).astype(np.int64)
        return dict(c_or_zero=(u == 0).sum(axis=2).astype(np.uint32), sum_or_union=np.ldexp(1.0, -u).sum(axis=2))
    est = np.zeros((len(ref), len(qry)))
    for i in range(len(ref)):
        for j in range(len(qry)):
            est[i, j] = lash_amd.ull_estimate(O.merge_images(O.ULL, p, ref[i], qry[j])[8:], p, estimator)
    return dict(sum_or_union=est)


def _genomes_for_dist():
    import oracle_lib as O
    rng = np.random.default_rng(12)
    base = O.synth_genome(600, 120_000)
    gs = [base]
    for i, rate in enumerate((0.001, 0.01, 0.05, 0.2)):
        g = base.copy()
        idx = rng.random(len(g)) < rate
        g[idx] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(idx.sum()))
        gs.append(g)
    gs += [O.synth_genome(601, 90_000), O.synth_genome(602, 150_001)]
    return gs


def _allpairs_worker(rank, world, port, algo, p, k, q):
    import oracle_lib as O
    from lash_amd.allpairs import all_vs_all
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gs = _genomes_for_dist()
        blocks = shard_genomes([len(g) for g in gs], world)
        a, b = blocks[rank]
        aid = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}[algo]
        imgs = [O.sketch_genomes(aid, k, p, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0] for g in gs[a:b]]
        local = torch.from_numpy(np.stack(imgs)) if imgs else torch.zeros((0, O.image_bytes(aid, p)), dtype=torch.uint8)
        bands = all_vs_all(algo, p, k, local, [e - s for s, e in blocks], pair_stats=_numpy_pair_stats, estimator="fgra")
        q.put((rank, bands))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,algo,p,k", [(2, "hmh", 0, 16), (3, "hmh", 0, 21), (2, "hll", 10, 16), (2, "ull", 10, 16)])
def test_all_vs_all_rows_over_ranks_equal_single_process(world, algo, p, k):
    import lash_amd
    import oracle_lib as O
    gs = _genomes_for_dist()
    n = len(gs)
    aid = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}[algo]
    every = np.stack([O.sketch_genomes(aid, k, p, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0] for g in gs])
    card = np.array([lash_amd.sketch_cardinality(algo, p, im) for im in every])
    want = lash_amd.dist_rows(algo, p, k, 1, card, card, **_numpy_pair_stats(algo, p, "fgra", every, every))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allpairs_worker, args=(r, world, port, algo, p, k, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    # rank r owns bands r and 2W-1-r of 2W equal bands (the printed triangle is then balanced): together they partition [0, N)
    from lash_amd.allpairs import row_bands
    for r, bands in got:
        assert [(b0, b1) for b0, b1, _ in bands] == row_bands(n, r, world)
    got = sorted(b for _, bands in got for b in bands)
    assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got[:-1], got[1:]))
    full = np.concatenate([g[2] for g in got])
    assert full.shape == (n, n) and np.array_equal(full, want)
    if algo == "hmh":                                            # and the numbers are the reference's formula (pure-Python restatement)
        import pyref as R
        for i in (0, 3, n - 1):
            for j in range(n):
                d = R.mash_distance(R.hmh_similarity(every[j].tobytes(), every[i].tobytes()), k, 1, False)
                assert abs(full[i, j] - d) < 1e-9
    assert 0 < full[0, 1] < full[0, 2] < full[0, 3] < full[0, 4] <= 1.0
