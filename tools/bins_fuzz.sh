#!/bin/bash
# tools/bins_fuzz.sh [tag] — the randomized runners pinned to the binned register tables (UltraLogLog p = 18 .. 22): default planning, a budget so small that a
# call runs many genome groups (or plans the global table), bins of 2^14 registers; GPU box
OUT=gpurun_out/${1:-bins_fuzz}; mkdir -p $OUT
run() { local name=$1; shift; env "$@" > $OUT/$name.log 2>&1; echo "$name rc=$? $(tail -1 $OUT/$name.log)"; }
run default      FUZZ_ALGO=ull FUZZ_P=18,19,20,21,22 timeout 1200 python3 tests/fuzz_gpu.py 300 $((71 + ${SEED_ADD:-0}))
run small_budget LASH_BINS_MB=200 FUZZ_ALGO=ull FUZZ_P=18,20,22 timeout 900 python3 tests/fuzz_gpu.py 150 $((72 + ${SEED_ADD:-0}))
run shift14      LASH_BIN_SHIFT=14 FUZZ_ALGO=ull FUZZ_P=18,20,22 timeout 900 python3 tests/fuzz_gpu.py 100 $((73 + ${SEED_ADD:-0}))
run raw          FUZZ_ALGO=ull FUZZ_P=18,22 timeout 900 python3 tests/fuzz_gpu_raw.py 100 $((74 + ${SEED_ADD:-0}))
run cli          FUZZ_ALGO=ull FUZZ_P=18,21 timeout 900 python3 tests/fuzz_gpu_cli.py 15 $((75 + ${SEED_ADD:-0}))
run layout       FUZZ_LAYOUT=kmer=lsb,codes=GATC FUZZ_ALGO=ull FUZZ_P=19,22 timeout 900 python3 tests/fuzz_gpu.py 100 $((76 + ${SEED_ADD:-0}))
