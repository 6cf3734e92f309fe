#!/bin/bash
# tools/profile_r02.sh <tag> <name> -- <bench.py args...>
# rocprofv3 passes over one bench.py workload on the GPU box (through gpurun, from the repo root): kernel trace + stats, then
# the PMC passes, each in its own run (never with sys/hip traces), the program directly after `--` (no wrapper).  The
# summaries land in gpurun_out/<tag>/<name>/ : kernel_stats.csv, pmc_summary.txt, bench.json (the line the same command prints
# without a profiler).  tools/profile_collect.py turns them into profiles/r02/ + profiles/traffic.json + profiles/valu.json.
set -u
TAG=$1; NAME=$2; shift 2; [ "${1:-}" = "--" ] && shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG/$NAME; mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py "$@" --no-cpu-baseline --no-parity-check --no-ubench > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py "$@" --no-cpu-baseline --no-parity-check --no-ubench > "$OUT/trace.log" 2>&1
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    N=$(echo $PMC | cut -d' ' -f1)
    timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 $REPO/bench.py "$@" --no-cpu-baseline --no-parity-check --no-ubench > "$OUT/pmc_$N.log" 2>&1
done
cd "$REPO"
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do grep -q sketch_kernel "$f" && cp "$f" "$OUT/kernel_stats.csv"; done
python3 tools/pmc_summary.py "$OUT" sketch_kernel pack_lookback finalize reduce_groups > "$OUT/pmc_summary.txt" 2>&1
find "$OUT" -name "*.db" -delete 2>/dev/null; find "$OUT" -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
rm -rf "$OUT"/pmc_*/ "$OUT"/trace 2>/dev/null
head -5 "$OUT/kernel_stats.csv"; head -12 "$OUT/pmc_summary.txt"
