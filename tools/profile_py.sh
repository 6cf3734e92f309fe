#!/bin/bash
# tools/profile_py.sh <tag> <name> <script.py> [args...] — rocprofv3 passes over a Python tool (GPU box, from the repo root): kernel trace +
# stats, then the PMC passes, each in a run of its own (never with sys / hip traces), python3 directly after `--`.
# -> gpurun_out/<tag>/<name>/ : kernel_stats.csv, pmc_summary.txt, run.log (the tool's own output without a profiler)
set -u
TAG=$1; NAME=$2; S=$3; shift 3
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG/$NAME; mkdir -p "$OUT"
export TMPDIR=/tmp
python3 $REPO/$S "$@" > "$OUT/run.log" 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/$S "$@" > "$OUT/trace.log" 2>&1
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    N=$(echo $PMC | cut -d' ' -f1)
    timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/$S "$@" > "$OUT/pmc_$N.log" 2>&1
done
cd "$REPO"
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats.csv"; done
python3 tools/pmc_summary.py "$OUT" sketch_kernel sole_ finalize census > "$OUT/pmc_summary.txt" 2>&1
find "$OUT" -name "*.db" -delete 2>/dev/null
rm -rf "$OUT"/pmc_*/ "$OUT"/trace 2>/dev/null
grep -E "sole_sketch|sketch_kernel" "$OUT/kernel_stats.csv" | head -3; head -12 "$OUT/pmc_summary.txt"
