#!/bin/bash
# tools/ab.sh — run bench.py against the default library and every build/variants/*.so (GPU box)
cd "$(dirname "$0")/.."
run() { LASH_GFX950_LIB=$1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline ${EXTRA:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s value %.4g  stages %s  %s' % (sys.argv[1], d['value'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()}, d.get('parity_vs_oracle')))" "$2"; }
run "" default
for f in build/variants/*.so; do run "$PWD/$f" "$(basename $f .so)"; done
