"""ctypes binding of liblash_gfx950.so — every symbol of include/lash_gfx950.h.

The library is the product; this module only declares prototypes.  It fails loudly when the shared object is
missing: there is no Python or CPU fallback for the hot path.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LASH_GFX950_LIB") or os.path.join(PKG, "liblash_gfx950.so")   # override: A/B builds

OK, EINVAL, ENODEV, EHIP, ENOMEM, ELIMIT, ERANGE, EFORMAT = 0, -1, -2, -3, -4, -5, -6, -7
HMH, HLL, ULL = 0, 1, 2
F_HMH_X_LOW, F_ACCUMULATE, F_NO_DIRECT, F_AMINO, F_STREAM_ONLY, F_NO_SOLE = 1, 2, 4, 8, 16, 32
FMT_FASTA, FMT_FASTQ = 1, 2
ABI_VERSION = 5


class Params(C.Structure):
    _fields_ = [("algo", C.c_int32), ("k", C.c_int32), ("p", C.c_int32), ("flags", C.c_uint32), ("seed", C.c_uint64)]


class Layout(C.Structure):
    """lash_layout: the reference's crate-internal rules as data."""
    _fields_ = [("base_code", C.c_uint8 * 4), ("kmer_lsb_first", C.c_uint8), ("hmh_x_low", C.c_uint8),
                ("hmh_reg_be", C.c_uint8), ("hll_bucket_high", C.c_uint8),
                ("hmh_header", C.c_char * 8), ("hll_header", C.c_char * 8), ("ull_header", C.c_char * 8),
                ("fastq_skip_bad", C.c_uint8), ("aa_code_zero_based", C.c_uint8), ("reserved", C.c_uint8 * 6)]


class Timing(C.Structure):
    _fields_ = [("pack_ms", C.c_float), ("sketch_ms", C.c_float), ("finalize_ms", C.c_float), ("calls", C.c_uint32),
                ("sketch_launches", C.c_uint32), ("sketch_workgroups", C.c_uint32), ("direct_launches", C.c_uint32),
                ("kmers", C.c_uint64), ("bases_last", C.c_uint64), ("packed_bytes", C.c_uint64),
                ("direct_ms", C.c_float), ("defer_launches", C.c_uint32), ("sole_launches", C.c_uint32)]


_vp, _u64, _u32, _int = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
_PP = C.POINTER(Params)
_LP = C.POINTER(Layout)

# name -> (restype, argtypes); must list every function include/lash_gfx950.h declares (tests check this)
PROTOTYPES = {
    "lash_abi_version": (_int, []),
    "lash_device_count": (_int, []),
    "lash_strerror": (C.c_char_p, [_int]),
    "lash_ctx_create": (_int, [C.POINTER(_vp), _int]),
    "lash_ctx_destroy": (None, [_vp]),
    "lash_ctx_set_stream": (_int, [_vp, _vp]),
    "lash_ctx_synchronize": (_int, [_vp]),
    "lash_ctx_last_error": (C.c_char_p, [_vp]),
    "lash_ctx_enable_timing": (_int, [_vp, _int]),
    "lash_ctx_get_timing": (_int, [_vp, C.POINTER(Timing)]),
    "lash_host_alloc_pinned": (_vp, [C.c_size_t]),
    "lash_host_free_pinned": (None, [_vp]),
    "lash_params_check": (_int, [_PP]),
    "lash_sketch_image_bytes": (C.c_size_t, [_int, _int]),
    "lash_layout_default": (None, [_LP]),
    "lash_layout_check": (_int, [_LP]),
    "lash_layout_parse": (_int, [C.c_char_p, _LP]),
    "lash_layout_header_bytes": (C.c_size_t, [_LP, _int]),
    "lash_layout_image_bytes": (C.c_size_t, [_LP, _int, _int]),
    "lash_ctx_set_layout": (_int, [_vp, _LP]),
    "lash_ctx_get_layout": (_int, [_vp, _LP]),
    "lash_sketch_batch": (_int, [_vp, _PP, _vp, _vp, _u64, _vp, _u32, _vp]),
    "lash_sketch_batch_async": (_int, [_vp, _PP, _vp, _vp, _u64, _vp, _u32, _vp]),
    "lash_sketch_batch_device": (_int, [_vp, _PP, _vp, _vp, _u64, _vp, _vp, _u32, _vp]),
    "lash_sketch_files_raw": (_int, [_vp, _PP, _vp, _vp, _vp, _u32, _vp]),
    "lash_sketch_files_raw_device": (_int, [_vp, _PP, _vp, _vp, _vp, _u32, _vp]),
    "lash_ctx_format_errors": (_u32, [_vp, _vp, _u32]),
    "lash_hll_replay_sums_device": (_int, [_vp, _PP, _vp, _vp, _u64, _vp, _u32, _vp]),
    "lash_hll_replay_streamed_chunk": (_int, [_vp, _PP, _vp, _u64, _int, _vp, _vp, _vp, _vp]),
    "lash_ctx_hll_inexact_sums": (_u32, [_vp, _vp, _u32]),
    "lash_fastq_valid_prefix": (_u64, [_vp, _u64]),
    "lash_fastq_neutralise_tail": (None, [_vp, _u64]),
    "lash_fastq_sanitize": (_u64, [_vp, _u64, _int]),
    "lash_pack_device": (_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u32, C.POINTER(_vp)]),
    "lash_sketch_packed_device": (_int, [_vp, _PP, _vp, _vp]),
    "lash_packed_free": (None, [_vp, _vp]),
    "lash_packed_bytes": (_u64, [_vp]),
    "lash_merge_images_device": (_int, [_vp, _int, _int, _vp, _vp, _u64]),
    "lash_merge_images": (_int, [_vp, _int, _int, _vp, _vp, _u64]),
    "lash_hmh_pair_counts_device": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, _vp]),
    "lash_hmh_pair_counts": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, _vp]),
    "lash_hll_pair_union_stats_device": (_int, [_vp, _int, _vp, _u32, _vp, _u32, _vp, _vp]),
    "lash_hll_pair_union_stats": (_int, [_vp, _int, _vp, _u32, _vp, _u32, _vp, _vp]),
    "lash_ull_pair_union_estimates_device": (_int, [_vp, _int, _int, _vp, _u32, _vp, _u32, _vp]),
    "lash_ull_pair_union_estimates": (_int, [_vp, _int, _int, _vp, _u32, _vp, _u32, _vp]),
    "lash_ull_estimate": (C.c_double, [_vp, _int, _int]),
    "lash_hmh_cardinality": (C.c_double, [_vp, _int]),
    "lash_hll_bias_load": (_int, [C.c_char_p, C.POINTER(_vp)]),
    "lash_hll_bias_from_arrays": (_int, [C.POINTER(_vp), _int, _vp, _vp, _u32]),
    "lash_hll_bias_has": (_int, [_vp, _int]),
    "lash_hll_bias_free": (None, [_vp]),
    "lash_hll_cardinality": (_int, [_vp, _int, _vp, C.POINTER(C.c_double)]),
    "lash_dist_rows": (_int, [_int, _int, _int, _int, _int, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "lash_hmh_pair_expected_collisions": (_int, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "lash_sketch_set_create": (_int, [_vp, _int, _int, _vp, _u32, _vp, _u32, C.POINTER(_vp)]),
    "lash_sketch_set_create_device": (_int, [_vp, _int, _int, _vp, _u32, C.POINTER(_vp)]),
    "lash_sketch_set_free": (None, [_vp, _vp]),
    "lash_sketch_set_size": (_u32, [_vp]),
    "lash_sketch_set_cardinalities": (_int, [_vp, _vp, _int, _vp, _vp, C.POINTER(_u32)]),
    "lash_sketch_set_prepare": (_int, [_vp, _vp, _vp]),
    "lash_sketch_set_hmh_expected_collisions": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _vp, C.POINTER(_u64)]),
    "lash_sketch_set_pair_block": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _int, _int, _vp, _vp, _vp]),
    "lash_sketch_set_pair_block_device": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _int, _int, _vp, _vp, _vp]),
    "lash_synth_genomes_device": (_int, [_vp, _u64, _u32, _u64, _vp]),
}

_lib = None


def load():
    """Returns the loaded library (cached).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: when PyTorch is around (tests, bench.py) its bundled libamdhip64 must be the copy
    # that gets loaded, so import torch BEFORE dlopen()ing our library (same SONAME: the loader then shares it).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s is missing: build it with `python -m lash_amd.build` (needs hipcc). "
            "lash_amd has no CPU fallback for the sketching hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.lash_abi_version() != ABI_VERSION:
        raise RuntimeError("liblash_gfx950.so ABI version %d != binding %d" % (lib.lash_abi_version(), ABI_VERSION))
    _lib = lib
    return lib
