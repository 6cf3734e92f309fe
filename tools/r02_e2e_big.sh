#!/bin/bash
# 10 Gbp with -t 1 (sequential gzip reader) and -t 16 (parallel members): same sketch; then configs[4] at its stated 100 Gbp
THREADS_LIST="1 16" bash tools/e2e_reads_gz.sh 10 r02_e2e_gz_10
INFLATE_ONLY=0 bash tools/e2e_reads_gz.sh 100 r02_e2e_gz_100
