#!/bin/bash
# tools/trace_gaps.sh <out> <script.py> [args...] — the timeline of one Python tool (GPU box): every kernel and memory copy with start, end
# and the gap to what ran before it on the device (rocprofv3 --kernel-trace --memory-copy-trace; no counters in this run).
OUT=$1; shift
REPO=$(pwd); mkdir -p "$REPO/gpurun_out/$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace_gaps -- python3 "$REPO/$1" "${@:2}" > "$REPO/gpurun_out/$OUT/trace.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/trace_gaps/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob("/tmp/trace_gaps/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
rows.sort()
t0 = rows[0][0] if rows else 0
prev = None
with open("gpurun_out/%s/timeline.txt" % sys.argv[1], "w") as o:
    for s, e, n in rows:
        o.write("%12.1f us  dur %10.1f us  gap %9.1f us  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0, n))
        prev = e
print(open("gpurun_out/%s/timeline.txt" % sys.argv[1]).read()[-6000:])
PY
