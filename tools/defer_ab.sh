#!/bin/bash
# tools/defer_ab.sh — HyperMinHash with the signature deferred against the plain kernel, by work-item size (GPU box).
# LASH_DEFER_MIN = bases per work item from which the deferring kernel is launched (-1: never, 0: always; default 2 000 000)
mkdir -p gpurun_out
run() { name=$1; shift; "$@" 2>&1 | tail -1 > gpurun_out/d_$name.json; python3 -c "
import json
d=json.load(open('gpurun_out/d_$name.json'))
print('$name', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], d.get('routes_agree'))
"; }
C="--no-cpu-baseline --no-ubench"
for df in -1 0; do
export LASH_DEFER_MIN=$df
run m${df}_L1M python bench.py --genomes 12000 --length 1000000 $C
run m${df}_L1500k python bench.py --genomes 8000 --length 1500000 $C
run m${df}_L2M python bench.py --genomes 6000 --length 2000000 $C
run m${df}_L3M python bench.py --genomes 4000 --length 3000000 $C
run m${df}_L5M python bench.py --genomes 2500 --length 5000000 $C
run m${df}_L12M python bench.py --genomes 1000 --length 12000000 $C
run m${df}_k21 python bench.py --genomes 2500 -k 21 $C
run m${df}_k11 python bench.py --genomes 2500 -k 11 $C
done
