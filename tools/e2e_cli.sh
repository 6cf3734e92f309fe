#!/bin/bash
# tools/e2e_cli.sh [n_genomes] — end-to-end `lash sketch` from FASTA files on tmpfs (host parse + PCIe + GPU + zstd),
# the number DESIGN.md quotes beside the HBM-resident rate.  Run on the GPU box.
# GZ=1: the files are gzip'd (level 6), the usual form of genome collections: the run is then inflate-bound on the host
# (compare with LASH_NO_FAST_INFLATE=1 = zlib only).
N=${1:-200}
REPO=$(pwd)
D=/dev/shm/lash_e2e_$$
mkdir -p $D && cd $D
python3 - <<PY
import sys
sys.path.insert(0, "$REPO")
import torch, lash_amd, zlib, os
from concurrent.futures import ThreadPoolExecutor
ctx = lash_amd.Context(0)
names = []
GZ = os.environ.get("GZ") == "1"
def gz_file(path):
    c = zlib.compressobj(6, zlib.DEFLATED, 31)                 # (zlib releases the GIL: threads compress in parallel)
    data = open(path, "rb").read()
    open(path + ".gz", "wb").write(c.compress(data) + c.flush())
    os.remove(path)
L = 5_000_000
for g0 in range(0, $N, 100):                               # the library's own generator, 100 genomes a time
    n = min(100, $N - g0)
    d = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    ctx.synth_genomes_device(g0, n, L, d)
    torch.cuda.synchronize()
    host = d.cpu().numpy()
    for g in range(g0, g0 + n):
        s = host[(g - g0) * L:(g - g0 + 1) * L].tobytes()
        with open("g%d.fa" % g, "wb") as f:
            f.write(b">g%d\n" % g)
            f.write(b"\n".join(s[i:i + 80] for i in range(0, len(s), 80)))
            f.write(b"\n")
        names.append("$D/g%d.fa" % g)
if GZ:
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(gz_file, names))
    names = [n + ".gz" for n in names]
open("list.txt", "w").write("\n".join(names) + "\n")
PY
for T in ${THREADS:-8 32 64}; do
  S=$(date +%s.%N); $REPO/lash_amd/bin/lash sketch -f list.txt -o out_$T -k 16 -t $T ${EXTRA:-} 2>&1 | tail -${TAILN:-1}; E=$(date +%s.%N)
  python3 -c "print(\"threads=$T wall=%.2f s  (start %.3f end %.3f)\" % ($E - $S, $S, $E))"
done
ls -la out_32_sketches.bin | awk '{print "sketches.bin bytes:", $5}'
cd / && rm -rf $D
