#!/bin/bash
# tools/asan_host.sh — the C++ host side (FASTX / gzip / zstd readers, parallel gzip, FASTQ validation, name order, JSON) built
# with AddressSanitizer + UBSan and driven through tests/test_host.py and tests/test_inflate.py (the DEFLATE decoder sees truncated and corrupted streams there).  CPU only (sanitizers are not available on the GPU pool;
# this file is listed in .gpurunignore).  The HIP library itself is linked as built: only the host objects are instrumented.
set -euo pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/lash_asan}; mkdir -p "$OUT"
H=$REPO/lash_amd/csrc/host
SRCS=$(python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from lash_amd.build import HOST_SOURCES
print(" ".join("$H/" + s for s in HOST_SOURCES if s != "main.cpp") + " $H/host_hooks.cpp")
PY
)
g++ -O1 -g -std=c++17 -fPIC -Wall -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o "$OUT/liblash_host.so" $SRCS \
    -L"$REPO/lash_amd" -llash_gfx950 -Wl,-rpath,"$REPO/lash_amd" -lz -ldl -lpthread
cd "$REPO"
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 LASH_HOST_LIB="$OUT/liblash_host.so" \
    python3 -m pytest tests/test_host.py tests/test_inflate.py -x -q -p no:cacheprovider "$@"
