#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. ..." — liblash_gfx950.so with sketch_kernels.hip compiled under extra flags, as
# build/variants/NAME.so (for A/B runs on one box: LASH_GFX950_LIB=$PWD/build/variants/NAME.so, tools/ab3.sh).
set -e
cd "$(dirname "$0")/.."
python3 -m lash_amd.build > /dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2 -c -o build/variants/$1.sketch_kernels.o lash_amd/csrc/sketch_kernels.hip
OBJS=""
for s in lash_api sketch_set pack_kernels fastq_check dist_kernels pair_planes dist_estimators; do OBJS="$OBJS build/obj/$s.hip.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$1.so $OBJS build/variants/$1.sketch_kernels.o
rm -f build/variants/$1.sketch_kernels.o
ls -la build/variants/$1.so
