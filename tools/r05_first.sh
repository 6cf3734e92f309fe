OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
