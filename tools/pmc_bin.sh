#!/bin/bash
# tools/pmc_bin.sh <tag> <kernel-substring> <binary> [args] — two SQ counter passes over a HIP binary (GPU box; rocprofv3 --pmc in runs of its own), summary per kernel
TAG=$1; KERN=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
B=$REPO/$1; shift
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_A" -- $B "$@" > "$OUT/A.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/pmc_B" -- $B "$@" > "$OUT/B.log" 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT $KERN
