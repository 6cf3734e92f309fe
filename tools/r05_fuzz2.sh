#!/bin/bash
# the randomized runners after LASH_SOLE_WGS went in (GPU box): the knob drawn per iteration, pinned to 1 / 2 / 3 workgroups, per algorithm
OUT=gpurun_out/r05_fuzz2; mkdir -p $OUT
n=0
for f in "X=1|fuzz_gpu.py 500 801" "X=1|fuzz_gpu.py 500 802" "X=1|fuzz_gpu.py 500 803" "X=1|fuzz_gpu_raw.py 300 804" "X=1|fuzz_gpu_raw.py 300 805" "X=1|fuzz_gpu_cli.py 40 806" "X=1|fuzz_gpu_stream.py 20 807" \
         "FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 500 811" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=2|fuzz_gpu.py 500 812" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=3|fuzz_gpu.py 500 813" \
         "FUZZ_SOLE=5000 FUZZ_SOLE_WGS=1|fuzz_gpu.py 500 814" "FUZZ_SOLE=700 FUZZ_SOLE_WGS=2|fuzz_gpu.py 500 815" \
         "FUZZ_ALGO=hll FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 400 821" "FUZZ_ALGO=ull FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 400 822" "FUZZ_ALGO=hmh FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 400 823" \
         "FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu_raw.py 300 831" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=2|fuzz_gpu_raw.py 300 832" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu_cli.py 40 833" \
         "FUZZ_SOLE=0|fuzz_gpu.py 400 841" "FUZZ_SOLE=0 LASH_STREAM_FIRST=1|fuzz_gpu.py 400 842" "FUZZ_SOLE=0 LASH_STREAM_FIRST=1 LASH_DEFER_MIN=0 FUZZ_ALGO=hmh|fuzz_gpu.py 400 843"; do
    E="${f%%|*}"; C="${f##*|}"; n=$((n+1))
    env $E timeout 1500 python3 tests/$C > $OUT/$n.log 2>&1; echo "$E $C rc=$? $(tail -1 $OUT/$n.log | cut -c1-250)"
done
