#!/bin/bash
TAG=${1:-r02_prof}
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_1000x5M -- --steps 20 --warmup 5
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_10000x5M -- --steps 5 --warmup 2 --genomes 10000
bash tools/profile_r02.sh $TAG dirty_nrun_hmh_k16 -- --steps 20 --warmup 5 --dirty nrun
bash tools/profile_r02.sh $TAG dirty_lower_hmh_k16 -- --steps 20 --warmup 5 --dirty lower
python3 bench.py --workload allpairs --genomes 1000 --steps 5 --warmup 1 > gpurun_out/$TAG/bench_allpairs_hmh.json 2> gpurun_out/$TAG/bench_allpairs_hmh.err; head -c 900 gpurun_out/$TAG/bench_allpairs_hmh.json; echo
python3 bench.py --workload allpairs --genomes 1000 --steps 5 --warmup 1 --algo ull -p 12 > gpurun_out/$TAG/bench_allpairs_ull.json 2> gpurun_out/$TAG/bench_allpairs_ull.err; head -c 600 gpurun_out/$TAG/bench_allpairs_ull.json; echo
timeout 900 python3 -m pytest tests/test_gpu_bench_launch.py -x -q 2>&1 | tail -5
python3 bench.py > gpurun_out/$TAG/bench_default_full.json 2> gpurun_out/$TAG/bench_default_full.err; head -c 300 gpurun_out/$TAG/bench_default_full.json; echo
