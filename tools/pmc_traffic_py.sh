#!/bin/bash
# tools/pmc_traffic_py.sh <tag> <script.py> [args] — HBM traffic counters (FETCH_SIZE, WRITE_SIZE; separate passes) of a Python tool's kernels (GPU box)
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
S=$1; shift
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_FETCH_SIZE" -- python3 $REPO/$S "$@" > "$OUT/F.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_WRITE_SIZE" -- python3 $REPO/$S "$@" > "$OUT/W.log" 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT ${KERNELS:-sketch_kernel bins_apply}
