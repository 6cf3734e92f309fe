#!/bin/bash
# tools/bench3.sh — the three judged shapes + the soft-masked line, one JSON summary line each (GPU box)
mkdir -p gpurun_out
python bench.py --genomes 10000 --algo hll -k 21 -p 14 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_hll.json
python bench.py --workload reads --algo ull -p 12 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_ull.json
python bench.py --genomes 2500 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_def.json
python bench.py --genomes 1000 --dirty lower --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b_lower.json
for f in hll ull def lower; do python3 -c "
import json,sys
d=json.load(open('gpurun_out/b_$f.json'))
print('$f', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'])
"; done
