#!/bin/bash
# tools/bins_ab.sh [tag] — binned register tables (ULL p = 18 .. 22) on the GPU box: parity of the tests that reach them, then the rates of
# 1 000 x 5 Mbp per knob setting (LASH_BIN_SHIFT=14: bins of 2^14 registers as in round 5; LASH_BINS_MB: HBM one group of genomes may take;
# LASH_BIN_ROWS_LOG2 / LASH_SKETCH_THREADS: staging rows per wave, waves per workgroup) and the kernels' split.
OUT=gpurun_out/${1:-bins_ab}; mkdir -p $OUT
K="rare or larger_than_lds or fallback or outgrow"
timeout 900 python3 -m pytest tests/test_gpu_rare_hashes.py tests/test_gpu_parity.py -q -m gpu -x -k "$K" > $OUT/pytest.log 2>&1; echo "parity: $(tail -1 $OUT/pytest.log)"
LASH_BIN_SHIFT=14 timeout 900 python3 -m pytest tests/test_gpu_rare_hashes.py tests/test_gpu_parity.py -q -m gpu -x -k "$K" > $OUT/pytest_shift14.log 2>&1; echo "parity LASH_BIN_SHIFT=14: $(tail -1 $OUT/pytest_shift14.log)"
LASH_BIN_ROWS_LOG2=6 timeout 900 python3 -m pytest tests/test_gpu_rare_hashes.py tests/test_gpu_parity.py -q -m gpu -x -k "$K" > $OUT/pytest_rows6.log 2>&1; echo "parity LASH_BIN_ROWS_LOG2=6: $(tail -1 $OUT/pytest_rows6.log)"
export SHAPES=${SHAPES:-ull:16:18,ull:16:19,ull:16:20,ull:16:21,ull:16:22}
for cfg in "" "LASH_BIN_SHIFT=14" "LASH_BINS_MB=6144" "LASH_BIN_SHIFT=14 LASH_BINS_MB=6144" "LASH_BIN_ROWS_LOG2=6" "LASH_BIN_ROWS_LOG2=6 LASH_SKETCH_THREADS=256" "LASH_SKETCH_THREADS=256" "LASH_SKETCH_THREADS=512"; do
    echo "== $cfg"; env $cfg python3 tools/large_tables_rate.py 2>&1 | grep "k-mers/s"
done | tee $OUT/rates.txt
for p in 18 20 22; do
    echo "== p=$p"; SHAPES=ull:16:$p bash tools/ktrace_py.sh tools/large_tables_rate.py 2>&1 | grep -E "sketch_kernel|bins_apply|fillBuffer|finalize"
done | tee $OUT/ktrace.txt
