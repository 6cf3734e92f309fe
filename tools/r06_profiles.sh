#!/bin/bash
# round 6: the judged single-GPU shapes again on the final tree (tools/profile_r02.sh each: bench line, rocprofv3 --kernel-trace --stats, FETCH / WRITE / SQ passes in runs
# of their own), the default shape under the x = low variant (the kernels VERDICT r5 asked to be made flip-ready), and the rate tools.  GPU box; gpurun_out/<tag>/.
TAG=${1:-r06_prof}
bash tools/profile_r02.sh $TAG default_hmh_k16_12500x5M -- --steps 5 --warmup 2
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_1000x5M -- --steps 20 --warmup 5 --genomes 1000
bash tools/profile_r02.sh $TAG cfg2_hll_p14_k21_10000x5M -- --steps 5 --warmup 2 --genomes 10000 --algo hll -p 14 -k 21
bash tools/profile_r02.sh $TAG cfg4shape_ull_p12_reads -- --steps 20 --warmup 5 --workload reads --algo ull -p 12 -k 16
bash tools/profile_r02.sh $TAG dirty_lower_hmh_k16 -- --steps 20 --warmup 5 --genomes 1000 --dirty lower
bash tools/profile_r02.sh $TAG layout_hmh_x_low_12500x5M -- --steps 5 --warmup 2 --layout hmh_x=low
bash tools/profile_r02.sh $TAG layout_hll_bucket_high_10000x5M -- --steps 5 --warmup 2 --genomes 10000 --algo hll -p 14 -k 21 --layout hll_bucket=high
python3 tools/layout_risk.py 1000 > gpurun_out/$TAG/layout_risk.txt 2>&1; cut -c1-120 gpurun_out/$TAG/layout_risk.txt
python3 tools/dirty_rate.py 1000 > gpurun_out/$TAG/dirty_rate.txt 2>&1
python3 tools/viral_rate.py > gpurun_out/$TAG/viral_rate.txt 2>&1; cat gpurun_out/$TAG/viral_rate.txt
python3 tools/small_genomes_rate.py > gpurun_out/$TAG/small_genomes_rate.txt 2>&1; tail -8 gpurun_out/$TAG/small_genomes_rate.txt
python3 tools/realistic_rate.py > gpurun_out/$TAG/realistic_rate.txt 2>&1; tail -8 gpurun_out/$TAG/realistic_rate.txt
python3 tools/reads_rate.py > gpurun_out/$TAG/reads_rate.txt 2>&1; tail -4 gpurun_out/$TAG/reads_rate.txt
python3 tools/large_tables_rate.py > gpurun_out/$TAG/large_tables_rate.txt 2>&1; tail -8 gpurun_out/$TAG/large_tables_rate.txt
$(pwd)/tools/ubench_hash > gpurun_out/$TAG/ubench_hash.txt 2>&1
python3 tools/box_info.py > gpurun_out/$TAG/box_info.txt 2>&1
python3 bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; tail -c 400 gpurun_out/$TAG/bench_default.json
du -sh gpurun_out/$TAG
