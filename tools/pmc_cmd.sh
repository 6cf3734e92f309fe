#!/bin/bash
# tools/pmc_cmd.sh <tag> <python script> [args] — SQ counter pass over a script (GPU box), summary per kernel
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
S=$1; shift
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_A" -- python3 $REPO/$S "$@" > "$OUT/A.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_B" -- python3 $REPO/$S "$@" > "$OUT/B.log" 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT ${KERNELS:-sketch_kernel}
