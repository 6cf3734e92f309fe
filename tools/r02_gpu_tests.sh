#!/bin/bash
# full GPU suite + default bench (regression check after kernel changes)
OUT=gpurun_out/${1:-r02_tests}; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 400 $OUT/bench_default.json; echo
