#!/bin/bash
# tools/defer_ablate.sh — timing-only ablations of the deferring kernel (build/variants/*.so from tools/variants.sh; GPU box)
C="--no-cpu-baseline --no-ubench --no-parity-check --genomes 2500"
for v in "" noappend noread g2; do
  if [ -n "$v" ]; then export LASH_GFX950_LIB=$PWD/build/variants/liblash_$v.so; fi
  python bench.py $C 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-full}', 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], d['roofline']['kernel'])"
done
