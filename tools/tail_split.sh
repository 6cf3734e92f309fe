#!/bin/bash
# tools/tail_split.sh — the tail of a launch: kernel and step time against batch size, with (default) and without (LASH_TAIL_SPLIT=1)
# the quartered last round (GPU box; profiles/r03/tail_split.txt)
mkdir -p gpurun_out
run() { name=$1; shift; "$@" 2>&1 | tail -1 > gpurun_out/g_$name.json; python3 -c "
import json
d=json.load(open('gpurun_out/g_$name.json'))
print('$name', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'])
"; }
C="--no-cpu-baseline --no-ubench"
for ts in 1 4; do
export LASH_TAIL_SPLIT=$ts
for g in 40 150 300 600 1200 2400; do run ts${ts}_g$g python bench.py --genomes $g $C; done
run ts${ts}_reads_ull python bench.py --workload reads --algo ull -p 12 $C
run ts${ts}_hll python bench.py --genomes 10000 --algo hll -k 21 -p 14 $C
run ts${ts}_default python bench.py $C
done
