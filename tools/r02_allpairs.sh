#!/bin/bash
OUT=gpurun_out/${1:-r02_allpairs}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_allpairs.py tests/test_gpu_dist.py -x -q > $OUT/pytest.log 2>&1
tail -30 $OUT/pytest.log
