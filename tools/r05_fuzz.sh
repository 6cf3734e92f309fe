# the randomized runners after the persistent small-genome kernel went in (GPU box): default knob draw, then FUZZ_SOLE=1 / small limits
OUT=gpurun_out/r05_fuzz; mkdir -p $OUT
for f in "fuzz_gpu.py 400 105" "fuzz_gpu_raw.py 250 109" "fuzz_gpu_cli.py 30 111" "fuzz_gpu_stream.py 20 107"; do
    set -- $f
    timeout 1500 python3 tests/$1 $2 $3 > $OUT/$1.log 2>&1; echo "$1 rc=$? $(tail -1 $OUT/$1.log)"
done
for f in "fuzz_gpu.py 400 205" "fuzz_gpu_raw.py 250 209" "fuzz_gpu_cli.py 30 211"; do
    set -- $f
    FUZZ_SOLE=1 timeout 1500 python3 tests/$1 $2 $3 > $OUT/sole1_$1.log 2>&1; echo "FUZZ_SOLE=1 $1 rc=$? $(tail -1 $OUT/sole1_$1.log)"
done
for f in "fuzz_gpu.py 300 305" "fuzz_gpu_raw.py 150 309"; do
    set -- $f
    FUZZ_SOLE=700 timeout 1500 python3 tests/$1 $2 $3 > $OUT/sole700_$1.log 2>&1; echo "FUZZ_SOLE=700 $1 rc=$? $(tail -1 $OUT/sole700_$1.log)"
done
