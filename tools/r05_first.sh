OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_hll_corner.py tests/test_gpu_cli.py tests/test_gpu_stream_gz.py -x -q -m gpu > $OUT/pytest_h.log 2>&1; tail -12 $OUT/pytest_h.log
timeout 600 python3 tests/fuzz_gpu_stream.py 20 41 | tail -1
