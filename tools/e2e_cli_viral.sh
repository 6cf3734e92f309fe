#!/bin/bash
# tools/e2e_cli_viral.sh [n_files] — `lash sketch` end to end on a viral-collection-shaped set of FASTA files (3..300 kbp, log-uniform, 1..4 records each,
# 80-column lines) on tmpfs -> sketches.bin; hmh and hll p=10, LASH_CLI_TIMING marks.  GPU box.
N=${1:-50000}
REPO=$(pwd)
D=/dev/shm/lash_e2e_viral_$$
mkdir -p $D && cd $D
python3 - <<PY
import os, numpy as np
rng = np.random.default_rng(13)
N = $N
lens = np.exp(rng.uniform(np.log(3e3), np.log(3e5), size=N)).astype(np.int64)
nrec = rng.integers(1, 5, size=N)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
big = acgt[rng.integers(0, 4, size=int(lens.max()) + 400000, dtype=np.uint8)]
names, total, kmers = [], 0, 0
for g in range(N):
    L, n = int(lens[g]), int(nrec[g])
    o = int(rng.integers(0, 400000))
    cuts = [0] + sorted(int(c) for c in rng.integers(1, L, size=n - 1)) + [L]
    parts = []
    for r in range(n):
        s = big[o + cuts[r]:o + cuts[r + 1]]
        kmers += max(len(s) - 15, 0)
        full = (len(s) // 80) * 80
        lines = np.empty((full // 80, 81), np.uint8)
        lines[:, :80] = s[:full].reshape(-1, 80)
        lines[:, 80] = 10
        parts.append(b">r%d\n" % r + lines.tobytes() + (s[full:].tobytes() + b"\n" if len(s) > full else b""))
    with open("v%d.fa" % g, "wb") as f:
        f.write(b"".join(parts))
    names.append("$D/v%d.fa" % g)
    total += L
open("list.txt", "w").write("\n".join(names) + "\n")
open("meta.txt", "w").write("%d %d\n" % (total, kmers))
PY
read TOTAL KMERS < meta.txt
for A in "hmh 0" "hll 10"; do
  set -- $A
  S=$(date +%s.%N); LASH_CLI_TIMING=1 $REPO/lash_amd/bin/lash sketch -f list.txt -o out_$1 -k 16 -a $1 $( [ $1 != hmh ] && echo -p $2 ) -t ${THREADS:-16} 2>&1 | grep -E "context ready|writer done|^sketched" ; E=$(date +%s.%N)
  python3 -c "print('== %s: wall %.2f s for %d files, %.2f GB of bases -> %.3g k-mers/s, %.2f GB/s' % ('$1', $E - $S, $N, $TOTAL / 1e9, $KMERS / ($E - $S), $TOTAL / 1e9 / ($E - $S)))"
done
cd /; rm -rf $D
