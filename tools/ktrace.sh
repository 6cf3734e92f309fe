#!/bin/bash
# tools/ktrace.sh <tag> [bench args] — rocprofv3 kernel trace of bench.py, per-kernel average durations (GPU box)
TAG=${1:-kt}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-check "$@" > "$OUT/trace.log" 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"][:90]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(n, []).append(d)
for n, v in agg.items():
    print("%-92s n=%3d avg %9.1f us  min %9.1f" % (n, len(v), sum(v) / len(v), min(v)))
# timeline of the last step
last = rows[-12:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%8.1f -> %8.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:70]))
PY
