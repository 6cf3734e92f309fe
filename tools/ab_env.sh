#!/bin/bash
# tools/ab_env.sh "ENV=.. ENV=.." "bench args" [more "env" "args" pairs...] — kernel ms of bench.py under different environments, interleaved twice
for rep in 1 2; do
  i=1
  while [ $i -le $# ]; do
    e="${!i}"; j=$((i+1)); a="${!j}"
    ms=$(env $e python bench.py $a --no-cpu-baseline --no-parity-check --no-ubench 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms kernel, %.3f ms/step, %.4g k-mers/s' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['value']))" 2>&1 | tail -1)
    echo "[$e] [$a] $ms"
    i=$((i+2))
  done
done
