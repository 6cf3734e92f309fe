#!/bin/bash
# tools/ktrace_py.sh <script.py> [args] — lash:: kernels of a Python tool under rocprofv3 --kernel-trace --stats (GPU box)
REPO=$(pwd); OUT=$REPO/gpurun_out/kt_py; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; S=$REPO/$1; shift; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $S "$@" > $OUT/log.txt 2>&1
cd $REPO; tail -3 $OUT/log.txt
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lash::' in r['Name'] or 'rocclr' in r['Name']: print("%-86s calls %5s avg %10.1f us total %8.1f ms" % (r['Name'][:86], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
