#!/bin/bash
# round 5: the judged single-GPU shapes (tools/profile_r02.sh each) and the small-genome shapes of the persistent kernel
# (tools/profile_py.sh on tools/one_shape.py / tools/viral_rate.py) under rocprofv3, plus the rate tools.  GPU box; gpurun_out/<tag>/.
TAG=${1:-r05_prof}
bash tools/profile_r02.sh $TAG default_hmh_k16_12500x5M -- --steps 5 --warmup 2
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_1000x5M -- --steps 20 --warmup 5 --genomes 1000
bash tools/profile_r02.sh $TAG cfg2_hll_p14_k21_10000x5M -- --steps 5 --warmup 2 --genomes 10000 --algo hll -p 14 -k 21
bash tools/profile_r02.sh $TAG cfg4shape_ull_p12_reads -- --steps 20 --warmup 5 --workload reads --algo ull -p 12 -k 16
bash tools/profile_r02.sh $TAG dirty_lower_hmh_k16 -- --steps 20 --warmup 5 --genomes 1000 --dirty lower
bash tools/profile_py.sh $TAG/small hmh_100000x10k tools/one_shape.py 100000 10000 hmh 16 0
bash tools/profile_py.sh $TAG/small hll_p10_100000x10k tools/one_shape.py 100000 10000 hll 21 10
bash tools/profile_py.sh $TAG/small hmh_1000000x1k tools/one_shape.py 1000000 1000 hmh 16 0
bash tools/profile_py.sh $TAG/small hll_p10_1000000x1k tools/one_shape.py 1000000 1000 hll 21 10
bash tools/profile_py.sh $TAG/small hmh_20000x50k tools/one_shape.py 20000 50000 hmh 16 0
bash tools/profile_py.sh $TAG/small viral_200000 tools/viral_rate.py 200000
python3 tools/small_genomes_rate.py > gpurun_out/$TAG/small/small_genomes_rate.txt 2>&1; cat gpurun_out/$TAG/small/small_genomes_rate.txt
LASH_SOLE_MAX=0 python3 tools/small_genomes_rate.py > gpurun_out/$TAG/small/small_genomes_rate_sliced_kernels.txt 2>&1
python3 tools/viral_rate.py > gpurun_out/$TAG/small/viral_rate.txt 2>&1; cat gpurun_out/$TAG/small/viral_rate.txt
LASH_SOLE_MAX=0 python3 tools/viral_rate.py > gpurun_out/$TAG/small/viral_rate_sliced_kernels.txt 2>&1
python3 tools/dirty_rate.py 1000 > gpurun_out/$TAG/dirty_rate.txt 2>&1; cat gpurun_out/$TAG/dirty_rate.txt
python3 tools/realistic_rate.py > gpurun_out/$TAG/realistic_rate.txt 2>&1; tail -8 gpurun_out/$TAG/realistic_rate.txt
python3 tools/reads_rate.py > gpurun_out/$TAG/reads_rate.txt 2>&1; tail -4 gpurun_out/$TAG/reads_rate.txt
python3 tools/large_tables_rate.py > gpurun_out/$TAG/large_tables_rate.txt 2>&1; tail -8 gpurun_out/$TAG/large_tables_rate.txt
$(pwd)/tools/ubench_hash > gpurun_out/$TAG/ubench_hash.txt 2>&1
python3 tools/box_info.py > gpurun_out/$TAG/box_info.txt 2>&1
python3 bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; tail -c 400 gpurun_out/$TAG/bench_default.json
du -sh gpurun_out/$TAG
