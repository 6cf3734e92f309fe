#!/bin/bash
# tools/build_variant_lib.sh NAME "-DFLAG ..." — liblash_gfx950.so with sketch_kernels.hip compiled under extra flags, as build/variants/liblash_NAME.so
# (LASH_GFX950_LIB=$PWD/build/variants/liblash_NAME.so for A/B runs on one box; build/ is git-ignored but travels with gpurun).  Known variants:
#     trace   -DLASH_ITEM_TRACE_BUILD   per-workgroup trace of direct sketch launches (LASH_ITEM_TRACE=<file>, tools/item_trace.py)
set -e
cd "$(dirname "$0")/.."
python3 -m lash_amd.build > /dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2 -Iinclude -c -o build/variants/$1.sketch_kernels.o lash_amd/csrc/sketch_kernels.hip
OBJS=""
for s in lash_api lash_plan lash_hll_replay lash_dist_api sketch_set sole_kernels pack_kernels fastq_check dist_kernels pair_planes dist_estimators; do OBJS="$OBJS build/obj/$s.hip.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/liblash_$1.so $OBJS build/variants/$1.sketch_kernels.o
rm -f build/variants/$1.sketch_kernels.o
ls -la build/variants/liblash_$1.so
