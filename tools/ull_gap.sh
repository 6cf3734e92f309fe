#!/bin/bash
# tools/ull_gap.sh — the tail of a launch: finalize with more slices per genome (GPU box)
mkdir -p gpurun_out
run() { name=$1; shift; "$@" 2>&1 | tail -1 > gpurun_out/g_$name.json; python3 -c "
import json
d=json.load(open('gpurun_out/g_$name.json'))
print('$name', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'])
"; }
C="--no-cpu-baseline --no-ubench"
for cfg in "1 32" "4 32" "4 8" "4 4" "4 2"; do
set -- $cfg
export LASH_TAIL_SPLIT=$1 LASH_GROUP_FROM=$2
t=s$1g$2
run ${t}_g40 python bench.py --genomes 40 $C
run ${t}_g150 python bench.py --genomes 150 $C
run ${t}_g300 python bench.py --genomes 300 $C
run ${t}_g600 python bench.py --genomes 600 $C
run ${t}_g1200 python bench.py --genomes 1200 $C
run ${t}_g2400 python bench.py --genomes 2400 $C
run ${t}_reads2M_ull python bench.py --workload reads --reads 2000000 --algo ull -p 12 $C
done
