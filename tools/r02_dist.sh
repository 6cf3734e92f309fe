#!/bin/bash
OUT=gpurun_out/${1:-r02_dist}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_cli.py -x -q > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
