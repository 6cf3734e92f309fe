#!/bin/bash
# tools/e2e_inflate_ab.sh [GBP] [tag] — configs[4] end to end with the host's own inflate (default) and with zlib only
# (LASH_NO_FAST_INFLATE=1): same sketch, wall time of each.  GPU box.
GBP=${1:-10}; TAG=${2:-e2e_inflate_ab}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
g++ -O2 -std=c++17 -o tools/inflate_bench tools/inflate_bench.cpp lash_amd/csrc/host/inflate_fast.cpp -lz 2> $OUT/build.err
W=${WORKDIR:-/dev/shm/lash_e2e}; mkdir -p $W
MEMBERS=$(python3 -c "print(max(64, int($GBP * 6.4)))")
python3 tools/make_reads_gz.py $W/reads.fastq.gz $GBP $MEMBERS 16 > $OUT/make.txt 2>&1 || { cat $OUT/make.txt; exit 1; }
cat $OUT/make.txt
head -c 400000000 $W/reads.fastq.gz > $W/one.gz 2>/dev/null
python3 - <<PY
# the first member alone (a complete gzip file) for the single-thread decoder bench
import zlib
b = open("$W/one.gz", "rb").read()
d = zlib.decompressobj(31); d.decompress(b)
open("$W/member0.gz", "wb").write(b[:len(b) - len(d.unused_data)])
PY
tools/inflate_bench $W/member0.gz 2 | tee $OUT/inflate_bench.txt
echo "$W/reads.fastq.gz" > $W/list.txt
cd $W
for MODE in fast zlib; do
  for T in ${THREADS_LIST:-16 4 1}; do
    [ $MODE = zlib ] && export LASH_NO_FAST_INFLATE=1 || unset LASH_NO_FAST_INFLATE
    S0=$(date +%s.%N)
    $REPO/lash_amd/bin/lash sketch -f list.txt -a ull -p 12 -k 16 -t $T -o e2e_$MODE > $OUT/${MODE}_t$T.out 2> $OUT/${MODE}_t$T.err
    S1=$(date +%s.%N)
    python3 -c "
import sys, hashlib
sys.path.insert(0, '$REPO/tests'); sys.path.insert(0, '$REPO')
import host_lib as H
print('$MODE -t $T: %.2f s  sketch sha256 %s' % ($S1 - $S0, hashlib.sha256(H.zstd_read('$W/e2e_${MODE}_sketches.bin')).hexdigest()[:16]))" | tee -a $OUT/summary.txt
  done
done
unset LASH_NO_FAST_INFLATE
cd $REPO; rm -rf $W
