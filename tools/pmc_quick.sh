#!/bin/bash
# tools/pmc_quick.sh <tag> — two SQ counter passes + GRBM on bench.py (sketch kernel diagnosis)
TAG=${1:-q}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check ${EXTRA:-}"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/pmc_A" -- $BENCH > "$OUT/A.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/pmc_B" -- $BENCH > "$OUT/B.log" 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_C" -- $BENCH > "$OUT/C.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d "$OUT/pmc_D" -- $BENCH > "$OUT/D.log" 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT ${KERNELS:-sketch_kernel}
