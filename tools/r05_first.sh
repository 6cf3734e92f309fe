OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_bench_launch.py -x -q -m gpu > $OUT/pytest_e.log 2>&1; tail -4 $OUT/pytest_e.log
timeout 900 python3 bench.py --workload cli --genomes 10000 --steps 3 --warmup 1 > $OUT/bench_cli.json 2> $OUT/bench_cli.err; cat $OUT/bench_cli.json | cut -c1-900
