"""ctypes binding of the CPU oracle (oracle/liblash_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HMH, HLL, ULL = 0, 1, 2
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("LASH_ORACLE_LIB") or os.path.join(_ROOT, "oracle", "liblash_oracle.so")   # override: sanitizer builds


class Params(C.Structure):
    _fields_ = [("algo", C.c_int), ("k", C.c_int), ("p", C.c_int), ("seed", C.c_uint64), ("hmh_x_is_low", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    lib.lash_or_xxh3_64_8b.restype = C.c_uint64
    lib.lash_or_xxh3_64_8b.argtypes = [C.c_uint64, C.c_uint64]
    lib.lash_or_xxh3_128_4b.restype = None
    lib.lash_or_xxh3_128_4b.argtypes = [C.c_uint32, C.c_uint64, u64p, u64p]
    lib.lash_or_filter_out_n.restype = C.c_size_t
    lib.lash_or_filter_out_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    lib.lash_or_mask_bits.restype = C.c_uint64
    lib.lash_or_mask_bits.argtypes = [C.c_uint64, C.c_int]
    lib.lash_or_record_kmers.restype = C.c_uint64
    lib.lash_or_record_kmers.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.lash_or_image_bytes.restype = C.c_size_t
    lib.lash_or_image_bytes.argtypes = [C.c_int, C.c_int]
    lib.lash_or_sketch_genome.restype = C.c_int
    lib.lash_or_sketch_genome.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.lash_or_sketch_genomes.restype = C.c_int
    lib.lash_or_sketch_genomes.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_int]
    lib.lash_or_merge_images.restype = C.c_int
    lib.lash_or_merge_images.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lash_or_synth_genome.restype = None
    lib.lash_or_synth_genome.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    return lib


lib = _load()


def xxh3_64_8b(v, seed):
    return int(lib.lash_or_xxh3_64_8b(v & (2**64 - 1), seed & (2**64 - 1)))


def xxh3_128_4b(w, seed):
    lo, hi = C.c_uint64(), C.c_uint64()
    lib.lash_or_xxh3_128_4b(w & 0xFFFFFFFF, seed & (2**64 - 1), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def filter_out_n(seq: bytes) -> bytes:
    src = np.frombuffer(seq, dtype=np.uint8)
    out = np.empty(len(seq) + 1, dtype=np.uint8)
    m = lib.lash_or_filter_out_n(src.ctypes.data, len(seq), out.ctypes.data)
    return out[:m].tobytes()


def mask_bits(v, k):
    return int(lib.lash_or_mask_bits(v, k))


def record_kmers(seq: bytes, k: int) -> np.ndarray:
    src = np.frombuffer(seq, dtype=np.uint8) if len(seq) else np.zeros(1, np.uint8)
    out = np.empty(max(len(seq), 1), dtype=np.uint64)
    n = lib.lash_or_record_kmers(src.ctypes.data, len(seq), k, out.ctypes.data)
    return out[:n].copy()


def image_bytes(algo, p):
    return int(lib.lash_or_image_bytes(algo, p))


def sketch_genomes(algo, k, p, seed, seq: np.ndarray, rec_off: np.ndarray, genome_rec_off: np.ndarray,
                   threads=1, hmh_x_is_low=0) -> np.ndarray:
    """Returns images[n_genomes, image_bytes] (uint8)."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    if seq.size == 0:
        seq = np.zeros(1, np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    genome_rec_off = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
    n_g = len(genome_rec_off) - 1
    prm = Params(algo, k, p, seed, hmh_x_is_low)
    ib = image_bytes(algo, p)
    images = np.zeros((n_g, ib), dtype=np.uint8)
    rc = lib.lash_or_sketch_genomes(C.byref(prm), seq.ctypes.data, rec_off.ctypes.data, genome_rec_off.ctypes.data,
                                    n_g, images.ctypes.data, threads)
    if rc != 0:
        raise ValueError("oracle rejected parameters")
    return images


def merge_images(algo, p, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.empty_like(a)
    rc = lib.lash_or_merge_images(algo, p, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("merge failed")
    return out


def synth_genome(g: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    lib.lash_or_synth_genome(g, n, out.ctypes.data)
    return out
