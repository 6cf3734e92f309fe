#!/bin/bash
# tools/build_trace_lib.sh — build/variants/liblash_trace.so: the sliced sketch kernels with their per-workgroup trace compiled in (see tools/build_variant_lib.sh).
#     LASH_GFX950_LIB=$PWD/build/variants/liblash_trace.so python3 tools/dirty_one_trace.py 2500000
exec "$(dirname "$0")/build_variant_lib.sh" trace "-DLASH_ITEM_TRACE_BUILD"
