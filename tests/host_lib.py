"""ctypes access to lash_amd/liblash_host.so (test hooks over the C++ host code: FASTX, JSON, zstd, list files)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("LASH_HOST_LIB") or os.path.join(ROOT, "lash_amd", "liblash_host.so")   # override: sanitizer builds
CLI = os.path.join(ROOT, "lash_amd", "bin", "lash")


def _load():
    if not os.path.exists(SO) or not os.path.exists(CLI):
        from lash_amd.build import build_host
        build_host()
    import lash_amd
    lash_amd.load()                     # fixes the HIP runtime load order (torch first), see lash_amd/_lib.py
    lib = C.CDLL(SO)
    lib.lash_host_free.argtypes = [C.c_void_p]
    lib.lash_host_read_fastx.restype = C.c_void_p
    lib.lash_host_read_fastx.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_uint64)]
    lib.lash_host_json_array.restype = C.c_void_p
    lib.lash_host_json_array.argtypes = [C.c_char_p, C.c_uint64]
    lib.lash_host_write_parameters.restype = C.c_void_p
    lib.lash_host_write_parameters.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64]
    lib.lash_host_read_list.restype = C.c_void_p
    lib.lash_host_read_list.argtypes = [C.c_char_p, C.POINTER(C.c_uint64)]
    lib.lash_host_zstd_write.restype = C.c_void_p
    lib.lash_host_zstd_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
    lib.lash_host_stream_find_cut.restype = C.c_uint64
    lib.lash_host_stream_find_cut.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.POINTER(C.c_uint64)]
    lib.lash_host_pgzip_read.restype = C.c_void_p
    lib.lash_host_pgzip_read.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.lash_host_zstd_read.restype = C.c_void_p
    lib.lash_host_zstd_read.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.lash_host_gunzip.restype = C.c_void_p
    lib.lash_host_gunzip.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.lash_host_gunzip_windowed.restype = C.c_void_p
    lib.lash_host_gunzip_windowed.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64)]
    lib.lash_host_crc32.restype = C.c_uint32
    lib.lash_host_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_uint64]
    lib.lash_host_xxh3_64.restype = C.c_uint64
    lib.lash_host_xxh3_64.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
    lib.lash_host_name_order.restype = C.c_uint64
    lib.lash_host_name_order.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    return lib


lib = _load()


def _take_str(p):
    s = C.string_at(p).decode()
    lib.lash_host_free(p)
    return s


def read_fastx(path):
    """-> list of record byte strings (raises ValueError with the library's message)"""
    seq, rec = C.c_void_p(), C.c_void_p()
    nb, nr = C.c_uint64(), C.c_uint64()
    err = lib.lash_host_read_fastx(path.encode(), C.byref(seq), C.byref(nb), C.byref(rec), C.byref(nr))
    if err:
        raise ValueError(_take_str(err))
    s = C.string_at(seq, nb.value)
    off = np.frombuffer(C.string_at(rec, (nr.value + 1) * 8), dtype=np.uint64)
    lib.lash_host_free(seq)
    lib.lash_host_free(rec)
    return [s[int(off[i]):int(off[i + 1])] for i in range(nr.value)]


def json_array(items):
    return _take_str(lib.lash_host_json_array(("\n".join(items)).encode(), len(items)))


def write_parameters(prefix, algorithm, k, precision, seed):
    err = lib.lash_host_write_parameters(prefix.encode(), algorithm.encode(), k, precision, seed)
    if err:
        raise ValueError(_take_str(err))


def read_list(path):
    n = C.c_uint64()
    p = lib.lash_host_read_list(path.encode(), C.byref(n))
    s = _take_str(p)
    if n.value == 2**64 - 1:
        raise ValueError(s)
    return s.split("\n")[:n.value]


def zstd_write(path, data: bytes, level=3, workers=1):
    buf = np.frombuffer(data, dtype=np.uint8)
    err = lib.lash_host_zstd_write(path.encode(), buf.ctypes.data if len(data) else None, len(data), level, workers)
    if err:
        raise ValueError(_take_str(err))


def zstd_read(path) -> bytes:
    out, n = C.c_void_p(), C.c_uint64()
    err = lib.lash_host_zstd_read(path.encode(), C.byref(out), C.byref(n))
    if err:
        raise ValueError(_take_str(err))
    b = C.string_at(out, n.value)
    lib.lash_host_free(out)
    return b


def stream_find_cut(buf: bytes, fmt: int):
    """(cut, carry) of the CLI's large-file streamer for one chunk; fmt 1 = FASTA, 2 = FASTQ."""
    carry = C.create_string_buffer(64)
    n = C.c_uint64()
    cut = lib.lash_host_stream_find_cut(buf, len(buf), fmt, carry, C.byref(n))
    return int(cut), carry.raw[:n.value]


def pgzip_read(path, threads, read_size=1 << 20):
    """(bytes, members served by workers, members inflated sequentially); raises ValueError with the reader's message"""
    out, n = C.c_void_p(), C.c_uint64()
    counts = (C.c_uint64 * 2)()
    err = lib.lash_host_pgzip_read(path.encode(), threads, read_size, C.byref(out), C.byref(n), counts)
    if err:
        msg = C.string_at(err).decode()
        lib.lash_host_free(err)
        raise ValueError(msg)
    data = C.string_at(out.value, n.value)
    lib.lash_host_free(out)
    return data, int(counts[0]), int(counts[1])


def xxh3_64(data: bytes, seed=0):
    return int(lib.lash_host_xxh3_64(data, len(data), seed))


def name_order(names, seed=93):
    """indices into `names` in the order the reference's seeded hashbrown map yields its keys"""
    out = (C.c_uint32 * max(len(names), 1))()
    n = lib.lash_host_name_order("\n".join(names).encode(), len(names), seed, out)
    return [int(out[i]) for i in range(n)]


def gunzip(data: bytes) -> bytes:
    """inflate_fast.hpp: every member of a gzip image (raises ValueError with the decoder's message)"""
    out, n = C.c_void_p(), C.c_uint64()
    err = lib.lash_host_gunzip(data, len(data), C.byref(out), C.byref(n))
    if err:
        raise ValueError(_take_str(err))
    res = C.string_at(out.value, n.value)
    lib.lash_host_free(out)
    return res


def gunzip_windowed(data: bytes, window=4096, piece=1000):
    """(bytes of the FIRST member, compressed bytes consumed) through the bounded-memory sequential reader"""
    out, n, used = C.c_void_p(), C.c_uint64(), C.c_uint64()
    err = lib.lash_host_gunzip_windowed(data, len(data), window, piece, C.byref(out), C.byref(n), C.byref(used))
    if err:
        raise ValueError(_take_str(err))
    res = C.string_at(out.value, n.value)
    lib.lash_host_free(out)
    return res, int(used.value)


def crc32(data: bytes, crc=0) -> int:
    return int(lib.lash_host_crc32(crc, data, len(data)))
