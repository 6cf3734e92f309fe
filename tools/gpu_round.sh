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
# round 6: the unpinned rules as kernel variants — one layout pinned per run (by default the runners draw one per iteration), every HyperMinHash launch deferring
for f in "hmh hmh_x=low 41" "hll hll_bucket=high 42" "ull kmer=lsb 43" "hmh codes=GATC,kmer=lsb,hll_bucket=high,hmh_x=low 44"; do
    set -- $f
    LASH_DEFER_MIN=0 FUZZ_ALGO=$1 FUZZ_LAYOUT=$2 timeout 1500 python3 tests/fuzz_gpu.py 200 $(($3 + ${SEED_ADD:-0})) > $OUT/layout_$3.log 2>&1; echo "FUZZ_LAYOUT=$2 FUZZ_ALGO=$1 rc=$? $(tail -1 $OUT/layout_$3.log)"
done
# the persistent kernel's stale-state debug build (tools/build_debug_stale.sh, built before the call: build/ travels with gpurun)
if [ -f build/variants/liblash_stale.so ]; then
    S=$PWD/build/variants/liblash_stale.so
    LASH_GFX950_LIB=$S timeout 900 python3 -m pytest tests/test_gpu_sole.py -q -m gpu -x > $OUT/stale_sole.log 2>&1; echo "stale-state build, tests/test_gpu_sole.py: $(tail -1 $OUT/stale_sole.log)"
    for f in "fuzz_gpu.py 300 51 1" "fuzz_gpu.py 300 52 3" "fuzz_gpu_raw.py 150 53 1"; do
        set -- $f
        LASH_GFX950_LIB=$S FUZZ_SOLE=1 FUZZ_SOLE_WGS=$4 timeout 900 python3 tests/$1 $2 $(($3 + ${SEED_ADD:-0})) > $OUT/stale_$1_$3.log 2>&1; echo "stale-state build FUZZ_SOLE=1 FUZZ_SOLE_WGS=$4 $1 rc=$? $(tail -1 $OUT/stale_$1_$3.log)"
    done
fi
# round 6: the runners pinned to the binned register tables (UltraLogLog p = 18 .. 22: default planning, many genome groups, bins of 2^14 registers)
SEED_ADD=${SEED_ADD:-0} bash tools/bins_fuzz.sh ${1:-gpu_round}/bins_fuzz
LASH_TEST_SOLE_EVERYWHERE=1 timeout 1500 python3 -m pytest tests -q -m gpu > $OUT/sole_everywhere.log 2>&1; echo "LASH_TEST_SOLE_EVERYWHERE=1: $(grep -E 'passed|failed' $OUT/sole_everywhere.log | tail -1)"
python3 tools/dist_rate.py > $OUT/dist_rate.txt 2>&1; cat $OUT/dist_rate.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
