#!/bin/bash
OUT=gpurun_out/${1:-r02_perf}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_layout.py tests/test_gpu_fullsize.py -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python3 tests/fuzz_gpu.py 300 33 > $OUT/fuzz.log 2>&1; tail -1 $OUT/fuzz.log
tools/ubench_hash > $OUT/ubench_hash.txt 2>&1; cat $OUT/ubench_hash.txt
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B > $OUT/bench_hmh.json 2> $OUT/bench_hmh.err
$B -k 21 > $OUT/bench_hmh_k21.json 2> $OUT/bench_hmh_k21.err
$B -k 11 > $OUT/bench_hmh_k11.json 2> $OUT/bench_hmh_k11.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "%.4e"%j['value'], "ms/step %.3f"%j['ms_per_step'], "direct ms %.3f"%j['roofline']['avg_launch_ms'], "valu frac", j['roofline_valu']['frac'], "packed-resident %.3e"%j['packed_resident_kmers_per_s_this_rank'])
PY
