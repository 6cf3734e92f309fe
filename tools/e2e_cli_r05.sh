#!/bin/bash
# tools/e2e_cli_r05.sh [n_files] [distinct] — `lash sketch` end to end at configs[1]/[2] size: n_files FASTA paths
# (default 10 000 x 5 Mbp, 80-column lines) on tmpfs -> sketches.bin, LASH_CLI_TIMING=1 marks.  `distinct` files are generated
# (default 1 000 = 5 GB of tmpfs) and hard-linked up to n_files: the pipeline reads every path in full from RAM either way,
# and 50 GB of text does not have to fit the box's /dev/shm.  GPU box.
N=${1:-10000}; DISTINCT=${2:-1000}
REPO=$(pwd)
D=/dev/shm/lash_e2e_$$
mkdir -p $D && cd $D
python3 - <<PY
import sys, os
sys.path.insert(0, "$REPO")
import numpy as np, torch, lash_amd
ctx = lash_amd.Context(0)
L, N, DISTINCT = 5_000_000, $N, min($DISTINCT, $N)
names = []
for g0 in range(0, DISTINCT, 100):
    n = min(100, DISTINCT - g0)
    d = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    ctx.synth_genomes_device(g0, n, L, d)
    torch.cuda.synchronize()
    host = d.cpu().numpy().reshape(n, L // 80, 80)
    for i in range(n):
        lines = np.empty((L // 80, 81), dtype=np.uint8)
        lines[:, :80] = host[i]
        lines[:, 80] = 10
        with open("g%d.fa" % (g0 + i), "wb") as f:
            f.write(b">g%d\n" % (g0 + i))
            f.write(lines.tobytes())
for g in range(N):
    if g >= DISTINCT:
        os.link("g%d.fa" % (g % DISTINCT), "g%d.fa" % g)
    names.append("$D/g%d.fa" % g)
open("list.txt", "w").write("\n".join(names) + "\n")
PY
free -g | head -2; df -h /dev/shm | tail -1; nproc
for T in ${THREADS:-16 32 64}; do
  S=$(date +%s.%N); LASH_CLI_TIMING=1 $REPO/lash_amd/bin/lash sketch -f list.txt -o out_$T -k ${K:-16} -a ${ALGO:-hmh} ${EXTRA:-} -t $T 2>&1 | tail -${TAILN:-12}; E=$(date +%s.%N)
  python3 -c "
n, L, k = $N, 5000000, ${K:-16}
w = $E - $S
print('threads=$T wall=%.2f s -> %.3g k-mers/s, %.2f GB/s of FASTA text' % (w, n * (L - k + 1) / w, n * (L + L // 80 + 8) / w / 1e9))"
done
ls -la out_32_sketches.bin 2>/dev/null | awk '{print "sketches.bin bytes:", $5}'
cd / && rm -rf $D
