#!/bin/bash
# tools/ktrace_cmd.sh <tag> <python script> [args] — rocprofv3 kernel trace of a script, per-kernel averages (GPU box)
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
S=$1; shift
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/$S "$@" > "$OUT/trace.log" 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    agg.setdefault(r["Kernel_Name"][:100], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in agg.items():
    print("%-102s n=%3d avg %9.1f us  min %9.1f" % (n, len(v), sum(v) / len(v), min(v)))
PY
