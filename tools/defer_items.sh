run() { name=$1; shift; "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], d['roofline']['kernel'])
"; }
C="--no-cpu-baseline --no-ubench --no-parity-check"
for g in 1000 600 300 100; do
run g${g}_default python bench.py --genomes $g $C
LASH_DEFER_MIN=0 run g${g}_defer_all python bench.py --genomes $g $C
LASH_SLICE_FACTOR=2 run g${g}_sf2 python bench.py --genomes $g $C
LASH_SLICE_FACTOR=2 LASH_DEFER_MIN=0 run g${g}_sf2_defer_all python bench.py --genomes $g $C
LASH_SLICE_FACTOR=1 LASH_DEFER_MIN=0 run g${g}_sf1_defer_all python bench.py --genomes $g $C
done
LASH_DEFER_MIN=0 run reads_hmh_defer python bench.py --workload reads --algo hmh $C
LASH_DEFER_MIN=-1 run reads_hmh_plain python bench.py --workload reads --algo hmh $C
LASH_DEFER_MIN=0 run L1M_defer python bench.py --genomes 12000 --length 1000000 $C
LASH_DEFER_MIN=-1 run L1M_plain python bench.py --genomes 12000 --length 1000000 $C
LASH_DEFER_MIN=0 run L500k_defer python bench.py --genomes 24000 --length 500000 $C
LASH_DEFER_MIN=-1 run L500k_plain python bench.py --genomes 24000 --length 500000 $C
