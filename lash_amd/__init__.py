"""lash_amd — MI355X (gfx950) implementation of lash's sketching hot path.

The product is lash_amd/liblash_gfx950.so (HIP kernels + C ABI, include/lash_gfx950.h) and the C++ host tools in
lash_amd/csrc/host.  This Python package is the thin binding used by tests/ and bench.py: it owns no algorithm and
has no CPU fallback.
"""
from ._lib import (EHIP, EINVAL, ELIMIT, ENODEV, ERANGE, ENOMEM, F_ACCUMULATE, F_AMINO, F_HMH_X_LOW, F_NO_DIRECT, F_STREAM_ONLY, F_NO_SOLE, HLL, HMH, OK, ULL, Layout, Params,
                   Timing, load)
from .sketch import (ALGOS, Context, HllBias, LashError, Packed, PinnedArray, header_bytes, image_bytes, params_check, parse_layout,
                     records_to_arrays, sketch_cardinality, dist_rows, ull_estimate, SketchSet)

__all__ = ["ALGOS", "Context", "HllBias", "LashError", "Layout", "Packed", "PinnedArray", "Params", "Timing", "header_bytes", "image_bytes", "params_check",
           "parse_layout", "ull_estimate", "sketch_cardinality", "dist_rows", "SketchSet",
           "records_to_arrays", "load", "HMH", "HLL", "ULL", "F_ACCUMULATE", "F_AMINO", "F_HMH_X_LOW", "F_NO_DIRECT", "F_STREAM_ONLY", "F_NO_SOLE", "OK", "EINVAL", "ENODEV",
           "EHIP", "ENOMEM", "ELIMIT", "ERANGE"]
