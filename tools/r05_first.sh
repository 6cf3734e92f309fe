OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_bench_launch.py -x -q -m gpu > $OUT/pytest_c.log 2>&1; tail -8 $OUT/pytest_c.log
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json
