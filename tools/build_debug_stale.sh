#!/bin/bash
# tools/build_debug_stale.sh — liblash_gfx950.so with the persistent small-genome kernel's stale-state checks compiled in
# (-DLASH_DEBUG_STALE, sole_kernels.hip: rings and table must be at rest when a genome begins, or the kernel traps) as
# build/variants/liblash_stale.so.  Run the small-genome tests and the randomized runners against it with
#     LASH_GFX950_LIB=$PWD/build/variants/liblash_stale.so python3 -m pytest tests/test_gpu_sole.py -m gpu -q
#     LASH_GFX950_LIB=$PWD/build/variants/liblash_stale.so FUZZ_SOLE=1 FUZZ_SOLE_WGS=1 python3 tests/fuzz_gpu.py 300 31
# (build/ is git-ignored but travels with gpurun).
set -e
cd "$(dirname "$0")/.."
python3 -m lash_amd.build > /dev/null
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DLASH_DEBUG_STALE -Iinclude -c -o build/variants/stale.sole_kernels.o lash_amd/csrc/sole_kernels.hip
OBJS=""
for s in lash_api lash_plan lash_hll_replay lash_dist_api sketch_set sketch_kernels pack_kernels fastq_check dist_kernels pair_planes dist_estimators; do OBJS="$OBJS build/obj/$s.hip.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/liblash_stale.so $OBJS build/variants/stale.sole_kernels.o
rm -f build/variants/stale.sole_kernels.o
ls -la build/variants/liblash_stale.so
