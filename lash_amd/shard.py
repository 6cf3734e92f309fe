"""Multi-GPU sharding of the sketch path.

One sketch per file and no cross-file state (utils.rs:450-509) => genomes shard across ranks in contiguous blocks of
the file list, balanced by bytes, so that rank order == file order == the order `collect()` keeps (utils.rs:509) and
`{o}_files.json` records (utils.rs:577-580).  `sketch` needs no data-path collective; only consumers that want every
image on every GPU (the all-vs-all `dist`, utils.rs:84-373) gather them — one all_gather over RCCL on GPU tensors
("nccl" backend) or gloo on CPU tensors (tests).
"""
import numpy as np


def shard_genomes(byte_lens, world_size):
    """Contiguous blocks [start, end) per rank with (nearly) equal total bytes.  Every genome is in exactly one block."""
    n = len(byte_lens)
    cum = np.concatenate([[0], np.cumsum(np.asarray(byte_lens, dtype=np.float64))])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        i = int(np.searchsorted(cum, target, side="left"))
        # choose the closer of the two neighbouring cut points, never move backwards
        if i > 0 and abs(cum[i - 1] - target) <= abs(cum[min(i, n)] - target):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def gather_images(local_images, counts, group=None):
    """All-gather per-rank image blocks [n_local, image_bytes] (torch uint8 tensors on one device type) into
    [sum(counts), image_bytes] in rank (== file) order.  `counts[r]` = genomes on rank r (from shard_genomes)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(counts) == world and local_images.shape[0] == counts[rank]
    ib = local_images.shape[1]
    cmax = max(counts) if counts else 0
    padded = torch.zeros((cmax, ib), dtype=torch.uint8, device=local_images.device)
    padded[:counts[rank]] = local_images
    out = torch.empty((world * cmax, ib), dtype=torch.uint8, device=local_images.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    parts = [out[r * cmax:r * cmax + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0) if parts else out[:0]


def merge_partial_images(local_images, merge_into, group=None):
    """One huge input cut into positional chunks over the ranks (BASELINE configs[4]): every rank holds
    PARTIAL sketches [n, image_bytes] of the same n inputs.  All ranks end with the union.

    The union is not always an elementwise max (UltraLogLog's packed registers, the HyperLogLog header), so instead of an
    all-reduce the partial images are all-gathered — world_size x n x image_bytes, a few MB — and folded in rank order with
    `merge_into(dst, src)`, the caller's binding of `lash_merge_images[_device]` (Sketch::union / HyperLogLog::union /
    UltraLogLog::merge, utils.rs:171,261,357).  max / OR are commutative and idempotent: every rank gets identical bytes."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n, ib = local_images.shape
    out = torch.empty((world * n, ib), dtype=torch.uint8, device=local_images.device)
    dist.all_gather_into_tensor(out, local_images.contiguous(), group=group)
    acc = out[:n].clone()
    for r in range(1, world):
        merge_into(acc, out[r * n:(r + 1) * n])
    return acc


def effective_cores():
    """host cores this process may really use: the minimum of the logical CPUs, the affinity mask and the cgroup CPU quota
    (the GPU boxes report 256 logical CPUs and a 16-CPU quota)"""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)
