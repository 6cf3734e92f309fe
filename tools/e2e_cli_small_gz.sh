#!/bin/bash
# tools/e2e_cli_small_gz.sh [n_files] [length] — as e2e_cli_small.sh with every file gzip-compressed (.fa.gz: how viral collections are kept).  GPU box.
N=${1:-30000}; L=${2:-10000}
REPO=$(pwd)
D=/dev/shm/lash_e2e_smallgz_$$
mkdir -p $D && cd $D
python3 - <<PY
import os, zlib, numpy as np
rng = np.random.default_rng(1)
N, L = $N, $L
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
names = []
for g0 in range(0, N, 1000):
    n = min(1000, N - g0)
    host = acgt[rng.integers(0, 4, size=(n, L // 80, 80), dtype=np.uint8)]
    lines = np.empty((n, L // 80, 81), dtype=np.uint8)
    lines[:, :, :80] = host
    lines[:, :, 80] = 10
    for i in range(n):
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        with open("g%d.fa.gz" % (g0 + i), "wb") as f:
            f.write(co.compress(b">g%d\n" % (g0 + i) + lines[i].tobytes()) + co.flush())
        names.append("$D/g%d.fa.gz" % (g0 + i))
open("list.txt", "w").write("\n".join(names) + "\n")
PY
for T in ${THREADS:-16}; do
  S=$(date +%s.%N); LASH_CLI_TIMING=1 $REPO/lash_amd/bin/lash sketch -f list.txt -o out_$T -k 16 -a hll -p 10 -t $T 2>&1 | grep -v "batch planned\|batch read\|GPU done" | tail -6; E=$(date +%s.%N)
  python3 -c "print('== hll p=10 threads=%d: wall %.2f s for %d .fa.gz files of %d bp -> %.3g k-mers/s' % ($T, $E - $S, $N, $L, $N * ($L - 15) / ($E - $S)))"
done
cd /; rm -rf $D
