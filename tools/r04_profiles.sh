#!/bin/bash
# round 4: every judged single-GPU shape under rocprofv3 (tools/profile_r02.sh each: kernel trace + stats, then the PMC passes in
# runs of their own), the instruction-cost table (tools/ubench_isa), the amino-acid path with its lane-utilisation counters, the
# large-table rates, the HBM counter calibration, the kernels' own streams (tools/ubench_hash).  GPU box; results under gpurun_out/<tag>/.
TAG=${1:-r04_prof}
bash tools/profile_r02.sh $TAG default_hmh_k16_12500x5M -- --steps 5 --warmup 2
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_1000x5M -- --steps 20 --warmup 5 --genomes 1000
bash tools/profile_r02.sh $TAG cfg2_hll_p14_k21_10000x5M -- --steps 5 --warmup 2 --genomes 10000 --algo hll -p 14 -k 21
bash tools/profile_r02.sh $TAG cfg4shape_ull_p12_reads -- --steps 20 --warmup 5 --workload reads --algo ull -p 12 -k 16
bash tools/profile_r02.sh $TAG dirty_nrun_hmh_k16 -- --steps 20 --warmup 5 --genomes 1000 --dirty nrun
bash tools/profile_r02.sh $TAG dirty_lower_hmh_k16 -- --steps 20 --warmup 5 --genomes 1000 --dirty lower
REPO=$(pwd); export TMPDIR=/tmp
OUT=gpurun_out/$TAG/calibration; mkdir -p $OUT; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/cal_FETCH -- $REPO/tools/ubench hbm > $REPO/$OUT/cal_FETCH.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/cal_WRITE -- $REPO/tools/ubench hbm > $REPO/$OUT/cal_WRITE.log 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT > $OUT/calibration.txt 2>&1; rm -rf $OUT/cal_FETCH $OUT/cal_WRITE; head -30 $OUT/calibration.txt
$REPO/tools/ubench_hash > gpurun_out/$TAG/ubench_hash.txt 2>&1; cat gpurun_out/$TAG/ubench_hash.txt
mkdir -p gpurun_out/$TAG/isa_cost
$REPO/tools/ubench_isa --json gpurun_out/$TAG/isa_cost/costs_w4.json > gpurun_out/$TAG/isa_cost/ubench_isa_w4.txt 2>&1
$REPO/tools/ubench_isa --waves 2 > gpurun_out/$TAG/isa_cost/ubench_isa_w2.txt 2>&1
$REPO/tools/ubench_isa --waves 1 > gpurun_out/$TAG/isa_cost/ubench_isa_w1.txt 2>&1
# amino acids (f4): rate, kernel stats, lane utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
A=gpurun_out/$TAG/aa; mkdir -p $A
python3 tools/aa_rate.py > $A/aa_rate.txt 2>&1; cat $A/aa_rate.txt
cd /tmp
CHECK=0 STEPS=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$A/trace -- python3 $REPO/tools/aa_rate.py > $REPO/$A/trace.log 2>&1
CHECK=0 STEPS=3 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $REPO/$A/run/pmc -- python3 $REPO/tools/aa_rate.py > $REPO/$A/pmc.log 2>&1
cd $REPO
for f in $(find $A/trace -name "*kernel_stats.csv"); do cp $f $A/kernel_stats.csv; done
python3 tools/pmc_summary.py $A/run aa_sketch_kernel > $A/pmc_summary.txt 2>&1; cat $A/pmc_summary.txt
rm -rf $A/trace $A/run
# register tables beyond LDS: binned / byte tables (this build), and the global-atomic path for comparison
L=gpurun_out/$TAG/large_tables; mkdir -p $L
python3 tools/large_tables_rate.py > $L/rate_1000x5M.txt 2>&1; cat $L/rate_1000x5M.txt
SHAPES=ull:16:20,ull:16:18,hll:21:16 G=200 LASH_NO_BINS=1 LASH_NO_BYTES=1 python3 tools/large_tables_rate.py > $L/rate_global_atomics_200x5M.txt 2>&1; cat $L/rate_global_atomics_200x5M.txt
cd /tmp; G=400 SHAPES=ull:16:20 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$L/trace -- python3 $REPO/tools/large_tables_rate.py > $REPO/$L/trace.log 2>&1; cd $REPO
for f in $(find $L/trace -name "*kernel_stats.csv"); do cp $f $L/kernel_stats_ull_p20.csv; done; rm -rf $L/trace
python3 tools/dirty_rate.py 1000 > gpurun_out/$TAG/dirty_rate.txt 2>&1; cat gpurun_out/$TAG/dirty_rate.txt
python3 tools/realistic_rate.py > gpurun_out/$TAG/realistic_rate.txt 2>&1; cat gpurun_out/$TAG/realistic_rate.txt
python3 tools/reads_rate.py > gpurun_out/$TAG/reads_rate.txt 2>&1; cat gpurun_out/$TAG/reads_rate.txt
python3 tools/box_info.py > gpurun_out/$TAG/box_info.txt 2>&1
python3 bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; tail -c 600 gpurun_out/$TAG/bench_default.json
du -sh gpurun_out/$TAG
