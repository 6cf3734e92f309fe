#!/bin/bash
OUT=gpurun_out/${1:-r02_fuzz}; mkdir -p $OUT
for f in "fuzz_gpu.py 300 21" "fuzz_gpu_raw.py 300 5" "fuzz_gpu_cli.py 30 3" "fuzz_gpu_stream.py 20 3" "fuzz_gpu_dist.py 10 2"; do
    set -- $f
    timeout 1500 python3 tests/$1 $2 $3 > $OUT/$1.log 2>&1; echo "$1 rc=$? $(tail -1 $OUT/$1.log)"
done
