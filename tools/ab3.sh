#!/bin/bash
# tools/ab3.sh libA.so libB.so ... — kernel ms of the three judged shapes for each library, interleaved on one box
for rep in 1 2; do for lib in "$@"; do
  export LASH_GFX950_LIB=$PWD/$lib
  h=$(python bench.py --genomes 10000 --algo hll -k 21 -p 14 --no-cpu-baseline --no-parity-check --no-ubench 2>&1 | tail -1 | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['avg_launch_ms'])")
  u=$(python bench.py --workload reads --algo ull -p 12 --no-cpu-baseline --no-parity-check --no-ubench 2>&1 | tail -1 | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['avg_launch_ms'])")
  d=$(python bench.py --genomes 2500 --no-cpu-baseline --no-parity-check --no-ubench 2>&1 | tail -1 | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['roofline']['avg_launch_ms'])")
  echo "$lib hll $h ull $u def $d"
done; done
