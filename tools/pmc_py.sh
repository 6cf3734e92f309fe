#!/bin/bash
# tools/pmc_py.sh <script.py> "<PMC list>" [kernel-substring] — counters of a Python tool's kernels (rocprofv3 --pmc, its own run; GPU box)
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_py; rm -rf $OUT; mkdir -p $OUT/run; export TMPDIR=/tmp; S=$REPO/$1; PMC=$2; FILT=${3:-sketch_kernel}; cd /tmp
timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/run/pmc -- python3 $S > $OUT/log.txt 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT/run $FILT
