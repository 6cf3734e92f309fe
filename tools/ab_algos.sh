#!/bin/bash
# tools/ab_algos.sh — direct vs pack-first on the three sketch types (GPU box)
cd "$(dirname "$0")/.."
run() { timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g k-mers/s  %.3f ms/step  %s  %s' % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()}, d.get('parity_vs_oracle')))"; }
for cfg in "--algo hmh -k 16" "--algo hll -k 21 -p 14" "--algo ull -k 16 -p 12" "--algo hmh -k 31" "--algo hmh -k 11"; do
  echo "== $cfg"; echo -n "direct:     "; run $cfg; echo -n "pack-first: "; LASH_NO_DIRECT=1 run $cfg
done
