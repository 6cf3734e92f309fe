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
