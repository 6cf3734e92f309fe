#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/kt_contigs; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/contigs_rate.py 1000 > $OUT/log.txt 2>&1
cd $REPO; for f in $(find $OUT/trace -name "*kernel_stats.csv"); do python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lash::' in r['Name']: print("%-90s calls %5s avg %10.1f us total %8.1f ms" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
