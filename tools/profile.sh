#!/bin/bash
# tools/profile.sh <tag> — rocprofv3 passes for bench.py on the GPU box (run through gpurun from the repo root).
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, each in its own run (never with sys/hip traces).
# Summaries land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
    NAME=$(echo $PMC | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_$NAME" -- $BENCH > "$OUT/pmc_$NAME.log" 2>&1
done
# calibration of FETCH_SIZE / WRITE_SIZE on a known byte count (tools/ubench: 2 GiB copy and read, 16 B per lane)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/cal_FETCH" -- $REPO/tools/ubench hbm > "$OUT/cal_FETCH.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/cal_WRITE" -- $REPO/tools/ubench hbm > "$OUT/cal_WRITE.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
