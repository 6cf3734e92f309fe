#!/bin/bash
# in-place sparse-dirt handling: parity, fuzz, and the dirty / clean bench lines
OUT=gpurun_out/${1:-r02_dirt}; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_direct.py tests/test_gpu_parity.py tests/test_gpu_layout.py -x -q > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
timeout 1200 python3 tests/fuzz_gpu.py 400 7 > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for d in none nrun lower; do $B --dirty $d > $OUT/bench_$d.json 2> $OUT/bench_$d.err; done
$B --algo hll -p 14 -k 21 > $OUT/bench_hll.json 2> $OUT/bench_hll.err
$B --algo hll -p 14 -k 21 --dirty nrun > $OUT/bench_hll_nrun.json 2> $OUT/bench_hll_nrun.err
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r02_dirt*/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(os.path.dirname(f)), os.path.basename(f), "%.3e"%j['value'], "ms/step %.3f"%j['ms_per_step'], {k: round(v,3) for k,v in j['stage_ms_per_step'].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
