#!/bin/bash
OUT=gpurun_out/${1:-r02_dist2}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_allpairs.py tests/test_gpu_layout.py -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 600 python3 tests/fuzz_gpu_dist.py 10 5 > $OUT/fuzz_dist.log 2>&1; tail -1 $OUT/fuzz_dist.log
python3 tools/dist_rate.py > $OUT/dist_rate.txt 2>&1; cat $OUT/dist_rate.txt
