#!/bin/bash
# tools/tsan_zstd.sh — the parallel-frames zstd writer under ThreadSanitizer (CPU only; listed in .gpurunignore).
set -euo pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=${OUT:-/tmp/lash_tsan_zstd}; mkdir -p "$OUT"
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o "$OUT/tsan_zstd" "$REPO/tools/tsan_zstd.cpp" "$REPO/lash_amd/csrc/host/zstd_dl.cpp" -ldl
TSAN_OPTIONS="halt_on_error=0" "$OUT/tsan_zstd" "$OUT/x.bin" 2> "$OUT/tsan.log" | tail -1
echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' "$OUT/tsan.log" || true)"
