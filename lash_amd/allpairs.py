"""All-vs-all `dist` over several GPUs (BASELINE configs[3]).

The reference parallelises `dist` over reference sketches (`reference_sketches.par_iter()`, utils.rs:150,248,342) inside one
process.  Here: one process per GPU (`torch.distributed`, "nccl" = RCCL over xGMI).  Genomes shard across ranks in contiguous
byte-balanced blocks (shard.shard_genomes), every rank sketches its block, ONE collective follows — an all-gather of the
finished sketch images (32 KiB each for hmh: 100 000 genomes = 3.3 GB) — and rank r then owns reference rows
[r*N/W, (r+1)*N/W) of the distance matrix: pair statistics on its GPU (lash_*_pair_*_device), the O(pairs) arithmetic of the
reference's estimators on its host cores (lash_dist_rows), rows written by each rank into its own part file and
concatenated in rank order (= file order).  Nothing else crosses the links.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m lash_amd.allpairs -f list.txt -a hmh -k 16 -o dist.tsv

Row order is file order (`lash dist --file-order` prints the same); the reference's own order is its seeded hash map's key order,
which the C++ `lash dist` reproduces (lash_amd/csrc/host/name_order.cpp).
"""
import argparse
import gzip
import os
import sys

import numpy as np

from . import _lib
from .shard import gather_images, shard_genomes
from .sketch import ALGOS, Context, HllBias, dist_rows, header_bytes, image_bytes, sketch_cardinality


def row_block(n, rank, world):
    """reference rows owned by `rank`: [rank*n/world, (rank+1)*n/world)"""
    return n * rank // world, n * (rank + 1) // world


def cardinalities(algo, p, images, layout=None, estimator="fgra", hll_bias=None):
    return np.array([sketch_cardinality(algo, p, images[i], layout, estimator, hll_bias) for i in range(images.shape[0])], dtype=np.float64)


def gpu_pair_stats(ctx, algo, p, estimator, ref, qry):
    """pair statistics of ref[i] x qry[j] on the context's GPU.  ref / qry: numpy uint8 [n, image_bytes] (staged through the
    host-buffer entries) or CUDA torch tensors (device entries, nothing leaves HBM but the statistics)."""
    a = ALGOS[algo] if isinstance(algo, str) else int(algo)
    if isinstance(ref, np.ndarray):
        if a == _lib.HMH:
            c, n = ctx.hmh_pair_counts(ref, qry)
            return dict(c_or_zero=c, n_counts=n)
        if a == _lib.HLL:
            z, s = ctx.hll_pair_union_stats(p, ref, qry)
            return dict(c_or_zero=z, sum_or_union=s)
        return dict(sum_or_union=ctx.ull_pair_union_estimates(p, ref, qry, estimator))
    import torch
    nr, nq, dev = ref.shape[0], qry.shape[0], ref.device
    ref, qry = ref.contiguous(), qry.contiguous()
    torch.cuda.current_stream(dev).synchronize()              # the images come from torch's stream (the all-gather); the context runs on its own
    if a == _lib.HMH:
        c = torch.empty((nr, nq), dtype=torch.int32, device=dev)
        n = torch.empty_like(c)
        ctx.hmh_pair_counts_device(ref, nr, qry, nq, c, n)
        ctx.synchronize()
        return dict(c_or_zero=c.cpu().numpy().view(np.uint32), n_counts=n.cpu().numpy().view(np.uint32))
    if a == _lib.HLL:
        z = torch.empty((nr, nq), dtype=torch.int32, device=dev)
        s = torch.empty((nr, nq), dtype=torch.float64, device=dev)
        ctx.hll_pair_union_stats_device(p, ref, nr, qry, nq, z, s)
        ctx.synchronize()
        return dict(c_or_zero=z.cpu().numpy().view(np.uint32), sum_or_union=s.cpu().numpy())
    u = torch.empty((nr, nq), dtype=torch.float64, device=dev)
    ctx.ull_pair_union_estimates_device(p, estimator, ref, nr, qry, nq, u)
    ctx.synchronize()
    return dict(sum_or_union=u.cpu().numpy())


def all_vs_all(algo, p, k, local_images, counts, *, ctx=None, model=1, fp32=False, estimator="fgra", layout=None, group=None,
               pair_stats=None, max_block_pairs=1 << 25, hll_bias=None):
    """local_images: this rank's sketches, torch uint8 [counts[rank], image_bytes] (CUDA under nccl, CPU under gloo).
    Returns (r0, r1, dist) — float64 [r1 - r0, N]: this rank's rows of the distance matrix against ALL N sketches, before the
    "same name -> 0" rule.  pair_stats(algo, p, estimator, ref_block, all_images) -> dict of lash_dist_rows' arrays; default:
    the context's GPU kernels.  The one collective is gather_images()."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    every = gather_images(local_images, counts, group)                       # [N, image_bytes] on every rank, file order
    n = every.shape[0]
    r0, r1 = row_block(n, rank, world)
    host = every.cpu().numpy() if every.is_cuda else every.numpy()
    card = cardinalities(algo, p, host, layout, estimator, hll_bias)                # O(N * registers) host work, every rank the same
    if pair_stats is None:
        if ctx is None:
            raise ValueError("all_vs_all needs a lash_amd.Context (the pair kernels have no CPU fallback) or an explicit pair_stats")
        pair_stats = lambda a, pp, e, ref, qry: gpu_pair_stats(ctx, a, pp, e, ref, qry)   # noqa: E731
    out = np.zeros((r1 - r0, n), dtype=np.float64)
    step = max(1, max_block_pairs // max(n, 1))
    for b0 in range(r0, r1, step):
        b1 = min(r1, b0 + step)
        ref = every[b0:b1] if every.is_cuda else host[b0:b1]
        st = pair_stats(algo, p, estimator, ref, every if every.is_cuda else host)
        if ctx is not None and (ALGOS[algo] if isinstance(algo, str) else int(algo)) == ALGOS["hmh"]:
            # hyperminhash's expected collisions: the 65 536-cell regime (both sketches <= 2^19 distinct k-mers) on the GPU
            st["hmh_ec"] = ctx.hmh_pair_expected_collisions(card[b0:b1], card)
        out[b0 - r0:b1 - r0] = dist_rows(algo, p, k, model, card[b0:b1], card, fp32=fp32, hll_bias=hll_bias, **st)
    return r0, r1, out


def format_rows(names, r0, block, *, matrix=False, lower_triangle=True):
    """the text `lash dist` writes for reference rows r0.. (main.rs:436-466): TSV lines, or matrix rows ("\\n" + name + cells)"""
    lines = []
    for i in range(block.shape[0]):
        gi = r0 + i
        cells = []
        for j in range(block.shape[1]):
            if lower_triangle and j > gi:
                continue
            d = 0.0 if names[j] == names[gi] else block[i, j]
            cells.append((j, "%.6f" % d))
        if matrix:
            lines.append("\n" + names[gi] + "".join("\t" + c for _, c in cells))
        else:
            lines.extend("%s\t%s\t%s\n" % (names[gi], names[j], c) for j, c in cells)
    return "".join(lines)


def _read_plain(path):
    with open(path, "rb") as f:
        data = f.read()
    return gzip.decompress(data) if data[:2] == b"\x1f\x8b" else data


def main(argv=None):
    ap = argparse.ArgumentParser(description="sketch + all-vs-all dist over the ranks of a torch.distributed job")
    ap.add_argument("-f", "--file", required=True, help="list of FASTA/FASTQ files (plain or .gz), one per line")
    ap.add_argument("-o", "--output_file", default="dist")
    ap.add_argument("-a", "--algorithm", default="hmh")
    ap.add_argument("-k", "--kmer", type=int, default=16)
    ap.add_argument("-p", "--precision", type=int, default=10)
    ap.add_argument("-s", "--seed", type=int, default=42)
    ap.add_argument("-e", "--estimator", default="fgra")
    ap.add_argument("-m", "--model", type=int, default=1)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--dm", action="store_true")
    ap.add_argument("--backend", default=None, help="nccl (one GPU per rank, default) or gloo (ranks may share a GPU; images gathered on the host)")
    ap.add_argument("--device", type=int, default=None, help="GPU of this rank (default LOCAL_RANK; with --backend gloo several ranks may name the same one)")
    ap.add_argument("--hll-bias", default=os.environ.get("LASH_HLL_BIAS"), help="HLL++ bias tables (tools/ref_probe/extract_hll_bias.py); without them "
                    "hll estimates <= 5 * 2^p are refused")
    args = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = args.backend or "nccl"
    device = local if args.device is None else args.device
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    algo, p = args.algorithm, (args.precision if args.algorithm != "hmh" else 0)
    names = [ln for ln in open(args.file).read().split("\n") if ln.strip()]                       # main.rs:200-207
    sizes = [os.path.getsize(f) for f in names]
    blocks = shard_genomes(sizes, world)
    s, e = blocks[rank]
    ctx = Context(device)
    files = [_read_plain(f) for f in names[s:e]]
    imgs = ctx.sketch_files_raw(algo, args.kmer, p, args.seed, files) if files else np.zeros((0, image_bytes(algo, p)), np.uint8)
    local_images = torch.from_numpy(imgs)
    if backend == "nccl":
        local_images = local_images.cuda(device)
    r0, r1, block = all_vs_all(algo, p, args.kmer, local_images, [b - a for a, b in blocks], ctx=ctx, model=args.model, fp32=args.fp32,
                               estimator=args.estimator, hll_bias=HllBias(args.hll_bias) if args.hll_bias else None)
    part = "%s.part%d" % (args.output_file, rank)
    with open(part, "w") as f:
        f.write(format_rows(names, r0, block, matrix=args.dm))
    dist.barrier()
    if rank == 0:
        with open(args.output_file, "w") as out:
            out.write("".join("\t" + n for n in names) if args.dm else "Reference\tQuery\tDistance\n")   # main.rs:409-412, 439-441
            for r in range(world):
                with open("%s.part%d" % (args.output_file, r)) as f:
                    out.write(f.read())
                os.remove("%s.part%d" % (args.output_file, r))
        print("Distances computed.")
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
