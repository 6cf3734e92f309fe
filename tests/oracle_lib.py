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


class Layout(C.Structure):
    """lash_or_layout == the product's lash_layout (40 bytes): SURVEY App. D's unknowns U1-U6 as data."""
    _fields_ = [("base_code", C.c_uint8 * 4), ("kmer_lsb_first", C.c_uint8), ("hmh_x_low", C.c_uint8),
                ("hmh_reg_be", C.c_uint8), ("hll_bucket_high", C.c_uint8),
                ("hmh_header", C.c_char * 8), ("hll_header", C.c_char * 8), ("ull_header", C.c_char * 8),
                ("fastq_skip_bad", C.c_uint8), ("aa_code_zero_based", C.c_uint8), ("reserved", C.c_uint8 * 6)]

    def spec(self):
        """the text form lash_layout_parse() takes"""
        order = "".join("ACGT"[list(self.base_code).index(c)] for c in range(4))
        return ("codes=%s,kmer=%s,hmh_x=%s,hmh_reg=%s,hll_bucket=%s,hmh_hdr=%s,hll_hdr=%s,ull_hdr=%s,fastq_err=%s%s"
                % (order, "lsb" if self.kmer_lsb_first else "msb", "low" if self.hmh_x_low else "high",
                   "be" if self.hmh_reg_be else "le", "high" if self.hll_bucket_high else "low",
                   self.hmh_header.decode(), self.hll_header.decode(), self.ull_header.decode(), "skip" if self.fastq_skip_bad else "stop",
                   ",aa_codes=zero" if self.aa_code_zero_based else ""))


def make_layout(codes="ACGT", kmer="msb", hmh_x="high", hmh_reg="le", hll_bucket="low", hmh_hdr="", hll_hdr="azspl",
                ull_hdr="l", fastq_err="stop", aa_codes="one"):
    """codes: the four letters in code order (code 0 first), e.g. "ACGT" (kmerutils hypothesis) or "ACTG"."""
    lay = Layout()
    for code, letter in enumerate(codes):
        lay.base_code["ACGT".index(letter)] = code
    lay.kmer_lsb_first = int(kmer == "lsb")
    lay.hmh_x_low = int(hmh_x == "low")
    lay.hmh_reg_be = int(hmh_reg == "be")
    lay.hll_bucket_high = int(hll_bucket == "high")
    lay.hmh_header, lay.hll_header, lay.ull_header = hmh_hdr.encode(), hll_hdr.encode(), ull_hdr.encode()
    lay.fastq_skip_bad = int(fastq_err == "skip")
    lay.aa_code_zero_based = int(aa_codes == "zero")
    return lay


def parse_layout(spec):
    kw = {}
    for item in filter(None, (spec or "").split(",")):
        key, _, val = item.partition("=")
        kw[key.strip()] = val.strip()
    return make_layout(**kw)


class Params(C.Structure):
    _fields_ = [("algo", C.c_int), ("k", C.c_int), ("p", C.c_int), ("seed", C.c_uint64), ("hmh_x_is_low", C.c_int),
                ("layout", C.POINTER(Layout)), ("amino", C.c_int)]


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
    lib.lash_or_image_bytes_layout.restype = C.c_size_t
    lib.lash_or_image_bytes_layout.argtypes = [C.POINTER(Layout), C.c_int, C.c_int]
    lib.lash_or_header_bytes.restype = C.c_size_t
    lib.lash_or_header_bytes.argtypes = [C.POINTER(Layout), C.c_int]
    lib.lash_or_layout_check.restype = C.c_int
    lib.lash_or_layout_check.argtypes = [C.POINTER(Layout)]
    lib.lash_or_layout_default.restype = None
    lib.lash_or_layout_default.argtypes = [C.POINTER(Layout)]
    lib.lash_or_merge_images_layout.restype = C.c_int
    lib.lash_or_merge_images_layout.argtypes = [C.POINTER(Layout), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lash_or_sketch_file_buffers.restype = C.c_int
    lib.lash_or_sketch_file_buffers.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
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


def image_bytes(algo, p, layout=None):
    return int(lib.lash_or_image_bytes_layout(C.byref(layout) if layout is not None else None, algo, p))


def header_bytes(algo, layout=None):
    return int(lib.lash_or_header_bytes(C.byref(layout) if layout is not None else None, algo))


def default_layout():
    lay = Layout()
    lib.lash_or_layout_default(C.byref(lay))
    return lay


def _params(algo, k, p, seed, hmh_x_is_low, layout, amino=False):
    return Params(algo, k, p, seed, hmh_x_is_low, C.pointer(layout) if layout is not None else None, 1 if amino else 0)


def sketch_files(algo, k, p, seed, files_bytes, threads=1, hmh_x_is_low=0, layout=None, amino=False) -> np.ndarray:
    """files_bytes: list of uncompressed FASTA/FASTQ file contents -> images[n_files, image_bytes]
    (the oracle's own needletail-like parse + the per-file closure, utils.rs:452-508)."""
    n = len(files_bytes)
    bufs = (C.c_char_p * max(n, 1))(*files_bytes)
    lens = (C.c_uint64 * max(n, 1))(*[len(f) for f in files_bytes])
    prm = _params(algo, k, p, seed, hmh_x_is_low, layout, amino)
    images = np.zeros((n, image_bytes(algo, p, layout)), dtype=np.uint8)
    rc = lib.lash_or_sketch_file_buffers(C.byref(prm), bufs, lens, n, images.ctypes.data, threads)
    if rc != 0:
        raise ValueError("oracle: %s" % ("not a FASTA/FASTQ file" if rc == -2 else "rejected parameters"))
    return images


def sketch_genomes(algo, k, p, seed, seq: np.ndarray, rec_off: np.ndarray, genome_rec_off: np.ndarray,
                   threads=1, hmh_x_is_low=0, layout=None, amino=False) -> np.ndarray:
    """Returns images[n_genomes, image_bytes] (uint8)."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    if seq.size == 0:
        seq = np.zeros(1, np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    genome_rec_off = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
    n_g = len(genome_rec_off) - 1
    prm = _params(algo, k, p, seed, hmh_x_is_low, layout, amino)
    ib = image_bytes(algo, p, layout)
    images = np.zeros((n_g, ib), dtype=np.uint8)
    rc = lib.lash_or_sketch_genomes(C.byref(prm), seq.ctypes.data, rec_off.ctypes.data, genome_rec_off.ctypes.data,
                                    n_g, images.ctypes.data, threads)
    if rc != 0:
        raise ValueError("oracle rejected parameters")
    return images


def merge_images(algo, p, a: np.ndarray, b: np.ndarray, layout=None) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.empty_like(a)
    rc = lib.lash_or_merge_images_layout(C.byref(layout) if layout is not None else None, algo, p, a.ctypes.data,
                                         b.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("merge failed")
    return out


def synth_genome(g: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    lib.lash_or_synth_genome(g, n, out.ctypes.data)
    return out
