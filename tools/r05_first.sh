OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_hll_corner.py tests/test_gpu_rawfiles.py tests/test_gpu_parity.py tests/test_gpu_sole.py tests/test_amino.py -x -q -m gpu > $OUT/pytest_b.log 2>&1; tail -8 $OUT/pytest_b.log
