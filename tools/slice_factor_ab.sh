run() { name=$1; shift; "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], d['roofline']['kernel'])
"; }
C="--no-cpu-baseline --no-ubench --no-parity-check"
for g in 40 150 300 600 1000 1500 2400; do
LASH_SLICE_FACTOR=4 run g${g}_sf4 python bench.py --genomes $g $C
run g${g}_new python bench.py --genomes $g $C
done
LASH_SLICE_FACTOR=4 run reads_hmh_sf4 python bench.py --workload reads --algo hmh $C
run reads_hmh_new python bench.py --workload reads --algo hmh $C
LASH_SLICE_FACTOR=4 run lower_sf4 python bench.py --genomes 1000 --dirty lower $C
run lower_new python bench.py --genomes 1000 --dirty lower $C
