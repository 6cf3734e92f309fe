OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sole.py tests/test_gpu_direct.py -x -q -m gpu > $OUT/pytest_d.log 2>&1; tail -4 $OUT/pytest_d.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --workload reads --algo ull -p 12 -k 16 --no-cpu-baseline > $OUT/bench_reads.json 2>$OUT/bench_reads.err; python3 -c "
import json;j=json.loads(open('$OUT/bench_reads.json').read().strip().splitlines()[-1]);print('ull reads',j['value'],j['ms_per_step'],j['roofline']['frac'],j['roofline']['avg_launch_ms'])"
timeout 600 python3 bench.py --steps 5 --warmup 2 --genomes 10000 --algo hll -p 14 -k 21 --no-cpu-baseline > $OUT/bench_hll.json 2>$OUT/bench_hll.err; python3 -c "
import json;j=json.loads(open('$OUT/bench_hll.json').read().strip().splitlines()[-1]);print('hll cfg2',j['value'],j['ms_per_step'],j['roofline']['frac'],j['roofline']['avg_launch_ms'])"
