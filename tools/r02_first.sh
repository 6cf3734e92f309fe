#!/bin/bash
# first GPU pass of round 2: host provenance, configs[4] at full size, dirty-input and other-shape bench lines
OUT=gpurun_out/r02_first; mkdir -p $OUT
python3 tools/box_info.py > $OUT/box_info.txt 2>&1
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B > $OUT/bench_clean.json 2> $OUT/bench_clean.err
$B --dirty nrun > $OUT/bench_nrun.json 2> $OUT/bench_nrun.err
$B --dirty lower > $OUT/bench_lower.json 2> $OUT/bench_lower.err
$B --workload reads --algo ull -p 12 -k 16 > $OUT/bench_reads.json 2> $OUT/bench_reads.err
$B --algo hll -p 14 -k 21 --genomes 10000 > $OUT/bench_hll_10k.json 2> $OUT/bench_hll_10k.err
$B --genomes 10000 > $OUT/bench_hmh_10k.json 2> $OUT/bench_hmh_10k.err
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -s -k config4_full > $OUT/config4_full.log 2>&1
tail -5 $OUT/config4_full.log
head -c 600 $OUT/bench_nrun.json; echo; head -c 600 $OUT/bench_lower.json; echo
