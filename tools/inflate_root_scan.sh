#!/bin/bash
# tools/inflate_root_scan.sh — MB/s of inflate_fast.cpp with 9..12 first-level table bits on this host's CPU, one thread, on
# (a) a 50 Mbp synthetic genome as 80-column FASTA, gzip level 6, (b) 150-bp FASTQ reads with uniform random qualities.
OUT=${1:-gpurun_out/inflate_root_scan}; mkdir -p $OUT; W=/dev/shm/lash_root_scan; mkdir -p $W
python3 - <<PY
import numpy as np, zlib
rng = np.random.default_rng(1)
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), 50_000_000).tobytes()
fa = b">g\n" + b"\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + b"\n"
c = zlib.compressobj(6, zlib.DEFLATED, 31); open("$W/fa6.gz", "wb").write(c.compress(fa) + c.flush())
n = 200000
q = rng.integers(33, 74, 150 * n, dtype=np.uint8).tobytes()
fq = b"".join(b"@r%d\n" % i + seq[i * 150:(i + 1) * 150] + b"\n+\n" + q[i * 150:(i + 1) * 150] + b"\n" for i in range(n))
c = zlib.compressobj(6, zlib.DEFLATED, 31); open("$W/fq6.gz", "wb").write(c.compress(fq) + c.flush())
PY
for R in 9 10 11 12; do
    g++ -O2 -std=c++17 -DLASH_INFLATE_LL_ROOT=$R -o $OUT/bench_r$R tools/inflate_bench.cpp lash_amd/csrc/host/inflate_fast.cpp -lz || exit 1
    for f in fa6 fq6; do echo "root $R $($OUT/bench_r$R $W/$f.gz 5)"; done
done | tee $OUT/root_scan.txt
rm -rf $W $OUT/bench_r*
