#!/bin/bash
OUT=gpurun_out/${1:-gpu_longfuzz}; mkdir -p $OUT
S=${2:-101}
for f in "fuzz_gpu.py 1500 $S" "fuzz_gpu_raw.py 800 $((S+1))" "fuzz_gpu_cli.py 100 $((S+2))" "fuzz_gpu_stream.py 60 $((S+3))" "fuzz_gpu_dist.py 30 $((S+4))"; do
    set -- $f
    timeout 2400 python3 tests/$1 $2 $3 > $OUT/$1.log 2>&1; echo "$1 rc=$? $(tail -1 $OUT/$1.log)"
done
# the byte register tables (hll p = 16, ull p = 15 .. 17: LdsByteQRegs) and the binned ones (ull p = 18 .. 20), every iteration
FUZZ_ALGO=ull FUZZ_P=15,16,17 timeout 1200 python3 tests/fuzz_gpu.py 300 $((S+5)) > $OUT/bytes_ull.log 2>&1; echo "ull p=15..17 rc=$? $(tail -1 $OUT/bytes_ull.log)"
FUZZ_ALGO=hll FUZZ_P=16 timeout 1200 python3 tests/fuzz_gpu.py 200 $((S+6)) > $OUT/bytes_hll.log 2>&1; echo "hll p=16 rc=$? $(tail -1 $OUT/bytes_hll.log)"
FUZZ_ALGO=ull FUZZ_P=18,19,20 timeout 1200 python3 tests/fuzz_gpu.py 150 $((S+7)) > $OUT/bins_ull.log 2>&1; echo "ull p=18..20 rc=$? $(tail -1 $OUT/bins_ull.log)"
