"""All-vs-all `dist` over several GPUs (BASELINE configs[3]).

The reference parallelises `dist` over reference sketches (`reference_sketches.par_iter()`, utils.rs:150,248,342) inside one
process.  Here: one process per GPU (`torch.distributed`, "nccl" = RCCL over xGMI).  Genomes shard across ranks in contiguous
byte-balanced blocks (shard.shard_genomes), every rank sketches its block, ONE collective follows — an all-gather of the
finished sketch images (32 KiB each for hmh: 100 000 genomes = 3.3 GB) — and every rank then holds all N sketches resident on its
GPU (lash_sketch_set: cardinalities from register histograms made on the GPU, bit planes / threshold bitmaps built once) and
owns a share of the reference rows.  Only the lower triangle is printed (utils.rs:158-160), so rows are handed out in 2W bands
of equal height and rank r owns bands r and 2W-1-r: every rank prints the same number of pairs.  A band is processed in row
blocks: pair statistics of rows [b0, b1) x columns [0, b1) on the GPU with the tiles above the diagonal skipped, the O(pairs)
estimator arithmetic and the text of the rows on the rank's host cores in C++ (liblash_host.so, the code `lash dist` runs),
written to one part file per band; the parts are concatenated in band order (= file order).  Nothing else crosses the links.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m lash_amd.allpairs -f list.txt -a hmh -k 16 -o dist.tsv

Row order is file order (`lash dist --file-order` prints the same); the reference's own order is its seeded hash map's key order,
which the C++ `lash dist` reproduces (lash_amd/csrc/host/name_order.cpp).
"""
import argparse
import ctypes as C
import gzip
import os
import sys
import time

import numpy as np

from . import _lib
from .shard import effective_cores, gather_images, shard_genomes
from .sketch import ALGOS, ULL_ESTIMATORS, Context, HllBias, LashError, PinnedArray, _bias_handle, dist_rows, image_bytes, sketch_cardinality


def row_block(n, rank, world):
    """contiguous share of n rows: [rank*n/world, (rank+1)*n/world)"""
    return n * rank // world, n * (rank + 1) // world


def row_bands(n, rank, world):
    """the reference rows `rank` owns when only the lower triangle is printed: bands rank and 2*world-1-rank of 2*world equal
    bands — row i prints i+1 pairs, so a low band and its mirror image together print as much as any other such pair"""
    nb = 2 * world
    lo, hi = row_block(n, rank, nb), row_block(n, nb - 1 - rank, nb)
    return [b for b in (lo, hi) if b[1] > b[0]]


def cardinalities(algo, p, images, layout=None, estimator="fgra", hll_bias=None):
    """host path (tests without a GPU): one lash_*_cardinality call per image"""
    return np.array([sketch_cardinality(algo, p, images[i], layout, estimator, hll_bias) for i in range(images.shape[0])], dtype=np.float64)


def all_vs_all(algo, p, k, local_images, counts, *, ctx=None, model=1, fp32=False, estimator="fgra", layout=None, group=None,
               pair_stats=None, hll_bias=None):
    """Dense form (small N; tests): returns [(b0, b1, dist)] for this rank's bands — dist float64 [b1 - b0, N]: the band's rows
    of the distance matrix against ALL N sketches, before the "same name -> 0" rule.  local_images: this rank's sketches, torch
    uint8 [counts[rank], image_bytes] (CUDA under nccl, CPU under gloo).  pair_stats(algo, p, estimator, ref_block, all_images)
    -> dict of lash_dist_rows' arrays replaces the GPU (CPU tests); default: a lash_sketch_set on the context's GPU.  The one
    collective is gather_images()."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    every = gather_images(local_images, counts, group)                       # [N, image_bytes] on every rank, file order
    n = every.shape[0]
    a = ALGOS[algo] if isinstance(algo, str) else int(algo)
    out = []
    if pair_stats is not None:
        host = every.cpu().numpy() if every.is_cuda else every.numpy()
        card = cardinalities(algo, p, host, layout, estimator, hll_bias)
        for b0, b1 in row_bands(n, rank, world):
            st = pair_stats(algo, p, estimator, host[b0:b1], host)
            out.append((b0, b1, dist_rows(algo, p, k, model, card[b0:b1], card, fp32=fp32, hll_bias=hll_bias, **st)))
        return out
    if ctx is None:
        raise ValueError("all_vs_all needs a lash_amd.Context (the pair kernels have no CPU fallback) or an explicit pair_stats")
    s = ctx.sketch_set(a, p, every if every.is_cuda else every.numpy())
    card = s.cardinalities(estimator, hll_bias)
    s.prepare()
    for b0, b1 in row_bands(n, rank, world):
        st = s.pair_block(b0, b1, estimator=estimator)
        if a == _lib.HMH:
            ec = s.hmh_expected_collisions(b0, b1)
            if ec is not None:
                st["hmh_ec"] = ec
        out.append((b0, b1, dist_rows(algo, p, k, model, card[b0:b1], card, fp32=fp32, hll_bias=hll_bias, **st)))
    s.free()
    return out


class _Formatter:
    """liblash_host.so's row formatter (host/dist_format.cpp: the code `lash dist` runs): names and cardinalities go over once"""

    def __init__(self, names, card):
        from ctypes import CDLL
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "liblash_host.so")
        if not os.path.exists(so):
            from .build import build_host
            build_host()
        _lib.load()
        self.lib = CDLL(so)
        self.lib.lash_host_formatter_create.restype = C.c_void_p
        self.lib.lash_host_formatter_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        self.lib.lash_host_formatter_block.restype = C.c_int64
        self.lib.lash_host_formatter_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
        self.lib.lash_host_formatter_error.restype = C.c_char_p
        self.lib.lash_host_formatter_error.argtypes = [C.c_void_p]
        self.lib.lash_host_formatter_free.argtypes = [C.c_void_p]
        n = len(names)
        arr = (C.c_char_p * n)(*[s.encode() for s in names])
        card = np.ascontiguousarray(card, dtype=np.float64)
        self.h = self.lib.lash_host_formatter_create(arr, card.ctypes.data, n, arr, card.ctypes.data, n)

    def block(self, a, p, k, model, fp32, hll_bias, b0, b1, st, ld, matrix, threads, fd):
        g = lambda key: None if st.get(key) is None else st[key].ctypes.data   # noqa: E731
        w = self.lib.lash_host_formatter_block(self.h, a, int(p or 0), int(k), int(model), 1 if fp32 else 0, _bias_handle(hll_bias), b0, b1, 1,
                                               g("c_or_zero"), g("n_counts"), g("sum_or_union"), g("hmh_ec"), ld, 1 if matrix else 0, threads, fd)
        if w < 0:
            raise LashError(_lib.ERANGE, self.lib.lash_host_formatter_error(self.h).decode())
        return w

    def close(self):
        if self.h:
            self.lib.lash_host_formatter_free(self.h)
            self.h = None


def all_vs_all_stream(algo, p, k, local_images, counts, names, out_prefix, *, ctx, model=1, fp32=False, estimator="fgra", group=None,
                      matrix=False, hll_bias=None, max_block_pairs=1 << 25, threads=None, stats=None):
    """The production form: this rank's bands of the lower triangle, streamed block by block from the GPU's pair tables through the
    C++ formatter into `<out_prefix>.band<i>` files (band i of 2 * world; concatenated in band order they are the body of `lash dist
    --file-order`).  Memory is bounded by max_block_pairs whatever N is.  Returns the list of (band index, path, bytes)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    t0 = time.perf_counter()
    every = gather_images(local_images, counts, group)                       # [N, image_bytes] on every rank, file order
    n = every.shape[0]
    a = ALGOS[algo] if isinstance(algo, str) else int(algo)
    if every.is_cuda:
        import torch
        torch.cuda.current_stream(every.device).synchronize()   # the images come from torch's stream (the all-gather); the context runs on its own
    s = ctx.sketch_set(a, p, every if every.is_cuda else every.numpy())
    t1 = time.perf_counter()
    card = s.cardinalities(estimator, hll_bias)
    s.prepare()
    t2 = time.perf_counter()
    threads = threads or max(1, 2 * effective_cores() // max(1, world))   # (a little oversubscription hides the ordered writes)
    fmt = _Formatter(names, card)
    cap = max(max_block_pairs, n)
    # the block's pair tables land in page-locked memory (the copy back runs at the link rate).  Two slots: a helper thread has the
    # GPU compute block b+1 while this one turns block b into text (the C calls release the interpreter lock)
    from concurrent.futures import ThreadPoolExecutor
    pins = [{"c": PinnedArray(cap * 4, np.uint32) if a != _lib.ULL else None, "n": PinnedArray(cap * 4, np.uint32) if a == _lib.HMH else None,
             "u": PinnedArray(cap * 8, np.float64) if a != _lib.HMH else None} for _ in range(2)]
    pin = [{k_: v.array for k_, v in ps.items() if v is not None} for ps in pins]
    nb = 2 * world
    blocks = []                                                  # (band, b0, b1): rows [b0, b1) x columns [0, b1), <= cap pair-table entries
    for band in (rank, nb - 1 - rank):
        b_lo, b_hi = row_block(n, band, nb)
        b0 = b_lo
        while b0 < b_hi:
            b1 = min(b_hi, b0 + max(1, cap // max(b0 + 1, 1)))
            while b1 > b0 + 1 and (b1 - b0) * b1 > cap:
                b1 -= 1
            blocks.append((band, b0, b1))
            b0 = b1
    gpu_s = [0.0]

    def gpu_block(i):
        _, b0, b1 = blocks[i]
        tg = time.perf_counter()
        st = s.pair_block(b0, b1, n_cols=b1, triangle=True, estimator=estimator, out=pin[i & 1])
        if a == _lib.HMH:
            ec = s.hmh_expected_collisions(b0, b1, n_cols=b1)
            if ec is not None:
                st["hmh_ec"] = ec
        gpu_s[0] += time.perf_counter() - tg
        return st
    parts, t_host, pairs = [], 0.0, 0
    fds = {}
    for band in (rank, nb - 1 - rank):
        path = os.devnull if out_prefix == os.devnull else "%s.band%d" % (out_prefix, band)     # (os.devnull: a timing / census run)
        fds[band] = [path, os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644), 0]
    with ThreadPoolExecutor(1) as ex:
        fut = ex.submit(gpu_block, 0) if blocks else None
        for i, (band, b0, b1) in enumerate(blocks):
            st = fut.result()
            fut = ex.submit(gpu_block, i + 1) if i + 1 < len(blocks) else None
            th = time.perf_counter()
            fds[band][2] += fmt.block(a, p, k, model, fp32, hll_bias, b0, b1, st, b1, matrix, threads, fds[band][1])
            t_host += time.perf_counter() - th
            pairs += sum(range(b0 + 1, b1 + 1))
    for band, (path, fd, written) in fds.items():
        os.close(fd)
        parts.append((band, path, written))
    t_gpu = gpu_s[0]
    fmt.close()
    s.free()
    del pin, pins
    if stats is not None:
        stats.update(gather_s=t1 - t0, prepare_s=t2 - t1, pair_blocks_s=t_gpu, host_rows_s=t_host, printed_pairs=pairs, threads=threads)
    return parts


def format_rows(names, r0, block, *, matrix=False, lower_triangle=True):
    """pure-Python twin of the C++ formatter (tests): the text `lash dist` writes for reference rows r0.. (main.rs:436-466)"""
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
    ap.add_argument("-t", "--threads", type=int, default=0, help="host threads of this rank for the row arithmetic and text (default: its share of the cores)")
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
    parts = all_vs_all_stream(algo, p, args.kmer, local_images, [b - a for a, b in blocks], names, args.output_file, ctx=ctx, model=args.model,
                              fp32=args.fp32, estimator=args.estimator, matrix=args.dm, hll_bias=HllBias(args.hll_bias) if args.hll_bias else None,
                              threads=args.threads or None)
    del parts
    dist.barrier()
    if rank == 0:
        with open(args.output_file, "wb") as out:
            out.write(("".join("\t" + n for n in names) if args.dm else "Reference\tQuery\tDistance\n").encode())   # main.rs:409-412, 439-441
            for band in range(2 * world):
                path = "%s.band%d" % (args.output_file, band)
                with open(path, "rb") as f:
                    while True:
                        chunk = f.read(1 << 24)
                        if not chunk:
                            break
                        out.write(chunk)
                os.remove(path)
        print("Distances computed.")
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
