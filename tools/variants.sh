#!/bin/bash
# tools/variants.sh — build A/B variants of liblash_gfx950.so into build/variants/ (they travel with gpurun).
# usage: tools/variants.sh name "-DFLAG ..." [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $FLAGS \
     -o build/variants/liblash_$NAME.so lash_amd/csrc/*.hip &
done
wait
ls -la build/variants
