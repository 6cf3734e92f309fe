#!/bin/bash
# tools/tsan_pgzip.sh — builds tools/tsan_pgzip.cpp with the reader's sources under -fsanitize=thread and runs it on a generated
# 60-member file (with false member headers inside stored members).  CPU only; listed in .gpurunignore.
set -euo pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=${OUT:-/tmp/lash_tsan_pgzip}; mkdir -p "$OUT"
python3 - "$OUT/multi.gz" <<'PY'
import gzip, sys, numpy as np
rng = np.random.default_rng(3)
magic = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\x03"
parts = []
for i in range(60):
    body = bytes(np.frombuffer(b"ACGTN\n@+I", np.uint8)[rng.integers(0, 9, size=int(rng.integers(1000, 600000)))])
    parts.append(gzip.compress(magic * (i % 5) + body + magic, int(rng.integers(0, 7))))
open(sys.argv[1], "wb").write(b"".join(parts))
PY
H=$REPO/lash_amd/csrc/host
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o "$OUT/tsan_pgzip" "$REPO/tools/tsan_pgzip.cpp" "$H/pgzip.cpp" "$H/inflate_fast.cpp" -lz
TSAN_OPTIONS="halt_on_error=0" "$OUT/tsan_pgzip" "$OUT/multi.gz" 2> "$OUT/tsan.log" | tail -1
echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' "$OUT/tsan.log" || true)"
