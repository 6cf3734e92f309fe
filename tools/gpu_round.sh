#!/bin/bash
# tools/gpu_round.sh [tag] — the checks run on the GPU box before a round is called done:
#   the whole `-m gpu` suite, every randomized runner with a fixed seed (+ $SEED_ADD for other draws), the dist pair-kernel rates, smoke().
# Results under gpurun_out/<tag>/ (scratch); copy what should be kept into profiles/.
OUT=gpurun_out/${1:-gpu_round}; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
for f in "fuzz_gpu.py 300 5" "fuzz_gpu_cli.py 30 11" "fuzz_gpu_stream.py 20 7" "fuzz_gpu_raw.py 200 9" "fuzz_gpu_dist.py 20 13"; do
    set -- $f
    timeout 1500 python3 tests/$1 $2 $(($3 + ${SEED_ADD:-0})) > $OUT/$1.log 2>&1; echo "$1 rc=$? $(tail -1 $OUT/$1.log)"
done
# the same runners with every HyperMinHash launch deferring its signatures (by default only batches of work items >= 0.6 Mbp do)
for f in "fuzz_gpu.py 300 21" "fuzz_gpu_raw.py 150 22" "fuzz_gpu_cli.py 20 23"; do
    set -- $f
    LASH_DEFER_MIN=0 FUZZ_ALGO=hmh timeout 1500 python3 tests/$1 $2 $(($3 + ${SEED_ADD:-0})) > $OUT/defer_$1.log 2>&1; echo "LASH_DEFER_MIN=0 $1 rc=$? $(tail -1 $OUT/defer_$1.log)"
done
# the persistent small-genome kernel with SEVERAL genomes per workgroup (round 5: two bugs only that situation shows; tests/fuzz_knobs.py)
for f in "fuzz_gpu.py 300 31" "fuzz_gpu_raw.py 150 32" "fuzz_gpu_cli.py 20 33"; do
    set -- $f
    FUZZ_SOLE=1 FUZZ_SOLE_WGS=1 timeout 1500 python3 tests/$1 $2 $(($3 + ${SEED_ADD:-0})) > $OUT/wgs1_$1.log 2>&1; echo "FUZZ_SOLE=1 FUZZ_SOLE_WGS=1 $1 rc=$? $(tail -1 $OUT/wgs1_$1.log)"
done
python3 tools/dist_rate.py > $OUT/dist_rate.txt 2>&1; cat $OUT/dist_rate.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
