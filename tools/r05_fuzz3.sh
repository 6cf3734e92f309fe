#!/bin/bash
# the randomized runners after LASH_SOLE_WGS went in (GPU box): the knob drawn per iteration, pinned to 1 / 2 / 3 workgroups, per algorithm
OUT=gpurun_out/r05_fuzz3; mkdir -p $OUT
n=0
for f in "X=1|fuzz_gpu.py 500 901" "X=1|fuzz_gpu.py 500 902" "X=1|fuzz_gpu.py 500 903" "X=1|fuzz_gpu_raw.py 300 904" "X=1|fuzz_gpu_raw.py 300 905" "X=1|fuzz_gpu_cli.py 40 906" "X=1|fuzz_gpu_stream.py 20 907" \
         "FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 500 911" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=2|fuzz_gpu.py 500 912" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=3|fuzz_gpu.py 500 913" \
         "FUZZ_SOLE=5000 FUZZ_SOLE_WGS=1|fuzz_gpu.py 500 914" "FUZZ_SOLE=700 FUZZ_SOLE_WGS=2|fuzz_gpu.py 500 915" \
         "FUZZ_ALGO=hll FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 400 921" "FUZZ_ALGO=ull FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 400 922" "FUZZ_ALGO=hmh FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu.py 400 923" \
         "FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu_raw.py 300 931" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=2|fuzz_gpu_raw.py 300 932" "FUZZ_SOLE=1 FUZZ_SOLE_WGS=1|fuzz_gpu_cli.py 40 933" \
         "FUZZ_SOLE=0|fuzz_gpu.py 400 941" "FUZZ_SOLE=0 LASH_STREAM_FIRST=1|fuzz_gpu.py 400 942" "FUZZ_SOLE=0 LASH_STREAM_FIRST=1 LASH_DEFER_MIN=0 FUZZ_ALGO=hmh|fuzz_gpu.py 400 943"; do
    E="${f%%|*}"; C="${f##*|}"; n=$((n+1))
    env $E timeout 1500 python3 tests/$C > $OUT/$n.log 2>&1; echo "$E $C rc=$? $(tail -1 $OUT/$n.log | cut -c1-250)"
done
