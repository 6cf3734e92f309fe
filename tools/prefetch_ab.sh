#!/bin/bash
# tools/prefetch_ab.sh — the judged shapes after a kernel change, one line each (GPU box)
mkdir -p gpurun_out
run() { name=$1; shift; "$@" 2>&1 | tail -1 > gpurun_out/p_$name.json; python3 -c "
import json
d=json.load(open('gpurun_out/p_$name.json'))
print('$name', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], d['roofline']['kernel'], d.get('routes_agree','')[:40])
"; }
C="--no-cpu-baseline --no-ubench"
LASH_DEFER_MIN=-1 run plain_g2500 python bench.py --genomes 2500 $C
run defer_g2500 python bench.py --genomes 2500 $C
run cfg1 python bench.py --genomes 1000 $C
run hll python bench.py --genomes 10000 --algo hll -k 21 -p 14 $C
run reads_ull python bench.py --workload reads --algo ull -p 12 $C
run k21 python bench.py --genomes 2500 -k 21 $C
run k11 python bench.py --genomes 2500 -k 11 $C
run lower python bench.py --genomes 1000 --dirty lower $C
run nrun python bench.py --genomes 1000 --dirty nrun $C
run default python bench.py $C
