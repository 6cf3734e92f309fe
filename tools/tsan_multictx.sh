#!/bin/bash
# tools/tsan_multictx.sh [threads] [rounds] — build the library's HOST code with -fsanitize=thread into a scratch directory and
# run tools/tsan_multictx.cpp against it.  Reports how many ThreadSanitizer warnings name a lash_* frame (must be 0).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TSAN_DIR:-/tmp/lash_tsan}; mkdir -p "$OUT"
cd "$REPO/lash_amd/csrc"
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=thread -o "$OUT/liblash_gfx950.so" \
    lash_api.hip sketch_kernels.hip pack_kernels.hip dist_kernels.hip dist_estimators.hip > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 2; }
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=thread -pthread -o "$OUT/tsan_multictx" "$REPO/tools/tsan_multictx.cpp" -L"$OUT" -llash_gfx950 -Wl,-rpath,"$OUT" >> "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 2; }
cd "$REPO"
LASH_TRACE_HOST=1 TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4" "$OUT/tsan_multictx" "${1:-6}" "${2:-40}" > "$OUT/run.out" 2> "$OUT/run.err"
RC=$?
tail -1 "$OUT/run.out"
TOTAL=$(grep -c "WARNING: ThreadSanitizer" "$OUT/run.err")
OURS=$(awk '/WARNING: ThreadSanitizer/{blk=""} {blk=blk"\n"$0} /^$/{ if (blk ~ /lash_|lash::/) n++; blk=""} END{print n+0}' "$OUT/run.err")
echo "ThreadSanitizer warnings: $TOTAL total, $OURS naming a lash frame (exit code $RC)"
[ "$OURS" = "0" ] && [ "$RC" = "0" -o "$RC" = "66" ]
