OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_sole.py -q -m gpu -x > $OUT/sole_tests.log 2>&1; tail -12 $OUT/sole_tests.log
timeout 600 python3 tools/small_genomes_rate.py > $OUT/small.txt 2>&1; cat $OUT/small.txt
timeout 600 python3 tools/viral_rate.py > $OUT/viral.txt 2>&1; cat $OUT/viral.txt
