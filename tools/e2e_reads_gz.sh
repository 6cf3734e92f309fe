#!/bin/bash
# tools/e2e_reads_gz.sh [GBP] — BASELINE configs[4] end to end through the command line: one FASTQ.gz of GBP giga-bases of
# 150-bp reads (concatenated gzip members), `lash sketch -a ull -p 12 -k 16`, streamed in chunks with on-device accumulation.
# Reference behaviour matched: needletail streams the file, one sketch per file (utils.rs:453-459).
# Reports wall time, inflate rate (text bytes / wall) and k-mers/s; results under gpurun_out/<tag>/ (copy to profiles/r02/).
GBP=${1:-10}; TAG=${2:-r02_e2e_gz}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
W=${WORKDIR:-/dev/shm/lash_e2e}; mkdir -p $W
MEMBERS=$(python3 -c "print(max(64, int($GBP * 6.4)))")      # ~330 MB of text per member
python3 tools/make_reads_gz.py $W/reads.fastq.gz $GBP $MEMBERS 16 > $OUT/make.txt 2>&1 || { cat $OUT/make.txt; exit 1; }
cat $OUT/make.txt
echo "$W/reads.fastq.gz" > $W/list.txt
cd $W
for T in ${THREADS_LIST:-16}; do
    S0=$(date +%s.%N)
    $REPO/lash_amd/bin/lash sketch -f list.txt -a ull -p 12 -k 16 -t $T -o e2e > $OUT/sketch_t$T.out 2> $OUT/sketch_t$T.err
    S1=$(date +%s.%N)
    echo "process wall $(python3 -c "print('%.2f' % ($S1 - $S0))") s" >> $OUT/sketch_t$T.err
    tail -3 $OUT/sketch_t$T.err
    python3 -c "
import sys, hashlib
sys.path.insert(0, '$REPO/tests'); sys.path.insert(0, '$REPO')
import host_lib as H
print('-t $T sketch sha256', hashlib.sha256(H.zstd_read('$W/e2e_sketches.bin')).hexdigest())" | tee -a $OUT/sketch_sha.txt
done
# every thread count must give the same sketch (the zstd frame depends on -t, so compare the decompressed images)
python3 - <<PY | tee $OUT/same_sketch.txt
import glob, hashlib, sys
sys.path.insert(0, "$REPO/tests"); sys.path.insert(0, "$REPO")
import host_lib as H
import shutil, os
print("sha256 of the decompressed sketch image:", hashlib.sha256(H.zstd_read("$W/e2e_sketches.bin")).hexdigest())
PY
# the same text inflated only (zlib, one thread): the ceiling of any single-stream .gz reader
[ "${INFLATE_ONLY:-1}" = "1" ] && python3 - <<PY > $OUT/inflate_only.txt
import time, zlib
t0=time.perf_counter(); n=0
d=zlib.decompressobj(31)
with open("$W/reads.fastq.gz","rb") as f:
    while True:
        b=f.read(1<<24)
        if not b: break
        while b:
            n+=len(d.decompress(b)); b=d.unused_data
            if d.eof: d=zlib.decompressobj(31)
            else: break
dt=time.perf_counter()-t0
print("python zlib inflate only: %.2f GB of text in %.1f s = %.2f GB/s"%(n/1e9,dt,n/1e9/dt))
PY
cat $OUT/inflate_only.txt 2>/dev/null
cd $REPO
python3 - <<PY | tee $OUT/summary.txt
import re
mk=open("$OUT/make.txt").read()
reads=int(re.search(r"reads (\d+)",mk).group(1)); text=int(re.search(r"text_bytes (\d+)",mk).group(1)); fb=int(re.search(r"file_bytes (\d+)",mk).group(1))
err=open("$OUT/sketch_t16.err").read()
print(open("$OUT/sketch_sha.txt").read().strip())
m=re.search(r"in ([0-9.]+) s on",err); wall=float(m.group(1)) if m else float("nan")
el=re.search(r"process wall (\S+) s",err)
print("configs[4] end to end: %d reads (%.1f Gbp), %.2f GB of FASTQ text in a %.2f GB .gz" % (reads, reads*150/1e9, text/1e9, fb/1e9))
print("lash sketch -a ull -p 12 -k 16 -t 16: %.1f s inside sketch_files (process wall %s) = %.2f GB/s of text inflated+parsed+sketched, %.3g k-mers/s" % (wall, el.group(1) if el else "?", text/1e9/wall, reads*135/wall))
PY
rm -rf $W
