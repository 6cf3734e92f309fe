#!/bin/bash
# tools/e2e_cli_small.sh [n_files] [length] — `lash sketch` end to end on a collection of SMALL genomes (viruses, plasmids, contigs: one FASTA file
# each, 80-column lines) on tmpfs -> sketches.bin, with LASH_CLI_TIMING=1 marks: where does a file's time go when the GPU needs 20 ns for it?  GPU box.
N=${1:-100000}; L=${2:-10000}
REPO=$(pwd)
D=/dev/shm/lash_e2e_small_$$
mkdir -p $D && cd $D
python3 - <<PY
import os, numpy as np
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
        with open("g%d.fa" % (g0 + i), "wb") as f:
            f.write(b">g%d\n" % (g0 + i))
            f.write(lines[i].tobytes())
        names.append("$D/g%d.fa" % (g0 + i))
open("list.txt", "w").write("\n".join(names) + "\n")
PY
nproc
for A in ${ALGOS:-"hmh 0" "hll 10"}; do
  set -- $A
  for T in ${THREADS:-16 64}; do
    S=$(date +%s.%N); LASH_CLI_TIMING=1 $REPO/lash_amd/bin/lash sketch -f list.txt -o out_$1_$T -k 16 -a $1 $( [ $1 != hmh ] && echo -p $2 ) -t $T 2>&1 | tail -${TAILN:-25}; E=$(date +%s.%N)
    python3 -c "print('== %s threads=%d: wall %.2f s for %d files of %d bp -> %.3g k-mers/s' % ('$1', $T, $E - $S, $N, $L, $N * ($L - 15) / ($E - $S)))"
    ls -la out_$1_${T}_sketches.bin | awk '{print "sketches.bin bytes:", $5}'
  done
done
cd /; rm -rf $D
