#!/bin/bash
OUT=gpurun_out/${1:-r02_round2}; mkdir -p $OUT
timeout 2400 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_rawfiles.py -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for f in "fuzz_gpu_cli.py 40 17" "fuzz_gpu_stream.py 20 9"; do set -- $f; timeout 1500 python3 tests/$1 $2 $3 > $OUT/$1.log 2>&1; echo "$1 rc=$? $(tail -1 $OUT/$1.log)"; done
