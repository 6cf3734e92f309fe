#!/bin/bash
# tools/build_trace_lib.sh — liblash_gfx950.so with the sliced sketch kernels' per-workgroup trace compiled in (-DLASH_ITEM_TRACE_BUILD) as
# build/variants/liblash_trace.so.  With it, LASH_ITEM_TRACE=<file> makes every direct sketch launch append when (100 MHz wall clock) and where
# (XCC / shader engine / CU) each of its workgroups ran; tools/item_trace.py reads the file.
#     LASH_GFX950_LIB=$PWD/build/variants/liblash_trace.so python3 tools/dirty_one_trace.py 2500000
set -e
cd "$(dirname "$0")/.."
python3 -m lash_amd.build > /dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DLASH_ITEM_TRACE_BUILD -Iinclude -c -o build/variants/trace.sketch_kernels.o lash_amd/csrc/sketch_kernels.hip
OBJS=""
for s in lash_api lash_plan lash_hll_replay lash_dist_api sketch_set sole_kernels pack_kernels fastq_check dist_kernels pair_planes dist_estimators; do OBJS="$OBJS build/obj/$s.hip.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/liblash_trace.so $OBJS build/variants/trace.sketch_kernels.o
rm -f build/variants/trace.sketch_kernels.o
ls -la build/variants/liblash_trace.so
