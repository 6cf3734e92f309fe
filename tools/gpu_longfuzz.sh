#!/bin/bash
OUT=gpurun_out/${1:-gpu_longfuzz}; mkdir -p $OUT
S=${2:-101}
for f in "fuzz_gpu.py 1500 $S" "fuzz_gpu_raw.py 800 $((S+1))" "fuzz_gpu_cli.py 100 $((S+2))" "fuzz_gpu_stream.py 60 $((S+3))" "fuzz_gpu_dist.py 30 $((S+4))"; do
    set -- $f
    timeout 2400 python3 tests/$1 $2 $3 > $OUT/$1.log 2>&1; echo "$1 rc=$? $(tail -1 $OUT/$1.log)"
done
