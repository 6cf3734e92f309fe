#!/bin/bash
# tools/build_variant_sole.sh NAME "-DFLAG ..." — as tools/build_variant_lib.sh, for sole_kernels.hip: build/variants/liblash_NAME.so
set -e
cd "$(dirname "$0")/.."
python3 -m lash_amd.build > /dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2 -Iinclude -c -o build/variants/$1.sole_kernels.o lash_amd/csrc/sole_kernels.hip
OBJS=""
for s in lash_api lash_plan lash_hll_replay lash_dist_api sketch_set sketch_kernels pack_kernels fastq_check dist_kernels pair_planes dist_estimators; do OBJS="$OBJS build/obj/$s.hip.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/liblash_$1.so $OBJS build/variants/$1.sole_kernels.o
rm -f build/variants/$1.sole_kernels.o
ls -la build/variants/liblash_$1.so
