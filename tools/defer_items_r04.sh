# tools/defer_items_r04.sh: where deferred signatures start to pay with the round-4 forms (threshold words + per-lane stacks):
# whole genomes of a given length, every launch deferring (LASH_DEFER_MIN=0) against none (-1).  On the GPU box.
run() { name=$1; shift; "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'kernel %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'])
"; }
C="--no-cpu-baseline --no-ubench --no-parity-check --steps 10 --warmup 2"
for spec in "60000 200000" "40000 300000" "30000 400000" "24000 500000" "20000 600000" "16000 750000" "12000 1000000" "6000 2000000"; do
    set -- $spec
    LASH_DEFER_MIN=0 run L$2_defer python bench.py --genomes $1 --length $2 $C
    LASH_DEFER_MIN=-1 run L$2_plain python bench.py --genomes $1 --length $2 $C
done
