#!/bin/bash
# round 3: every judged single-GPU shape (and the dirty lines) under rocprofv3: tools/profile_r02.sh each (kernel trace + stats,
# then the PMC passes in runs of their own), the dist-side kernels, the HBM counter calibration, the issue ceilings
TAG=${1:-r03_prof}
bash tools/profile_r02.sh $TAG default_hmh_k16_12500x5M -- --steps 5 --warmup 2
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_1000x5M -- --steps 20 --warmup 5 --genomes 1000
bash tools/profile_r02.sh $TAG cfg2_hll_p14_k21_10000x5M -- --steps 5 --warmup 2 --genomes 10000 --algo hll -p 14 -k 21
bash tools/profile_r02.sh $TAG cfg4shape_ull_p12_reads -- --steps 20 --warmup 5 --workload reads --algo ull -p 12 -k 16
bash tools/profile_r02.sh $TAG dirty_nrun_hmh_k16 -- --steps 20 --warmup 5 --genomes 1000 --dirty nrun
bash tools/profile_r02.sh $TAG dirty_lower_hmh_k16 -- --steps 20 --warmup 5 --genomes 1000 --dirty lower
OUT=gpurun_out/$TAG/calibration; mkdir -p $OUT; REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/cal_FETCH -- $REPO/tools/ubench hbm > $REPO/$OUT/cal_FETCH.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/cal_WRITE -- $REPO/tools/ubench hbm > $REPO/$OUT/cal_WRITE.log 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT > $OUT/calibration.txt 2>&1; rm -rf $OUT/cal_FETCH $OUT/cal_WRITE; head -30 $OUT/calibration.txt
$REPO/tools/ubench_hash > gpurun_out/$TAG/ubench_hash.txt 2>&1; cat gpurun_out/$TAG/ubench_hash.txt
$REPO/tools/ubench_ops > gpurun_out/$TAG/ubench_ops.txt 2>&1; tail -20 gpurun_out/$TAG/ubench_ops.txt
# dist side: pair kernels under --kernel-trace --stats, and the rates the tools print
D=gpurun_out/$TAG/dist_kernels; mkdir -p $D; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$D/trace -- python3 $REPO/tools/pairs_rate.py > $REPO/$D/pairs_rate.txt 2>&1
cd $REPO; for f in $(find $D/trace -name "*kernel_stats.csv"); do cp $f $D/kernel_stats.csv; done; rm -rf $D/trace
: > gpurun_out/$TAG/pairs_rate.txt
for cfg in "ALGO=hmh" "ALGO=hmh FULL=0" "ALGO=hmh LASH_HMH_PAIRS_WORDS=1" "ALGO=hll P=14" "ALGO=ull P=12" "ALGO=ull P=16 N=2048"; do
    echo "== $cfg" >> gpurun_out/$TAG/pairs_rate.txt
    ( export $cfg; python3 tools/pairs_rate.py 2>&1 | tail -6 ) >> gpurun_out/$TAG/pairs_rate.txt
done
cat gpurun_out/$TAG/pairs_rate.txt
python3 tools/dist_rate.py > gpurun_out/$TAG/dist_rate.txt 2>&1; tail -30 gpurun_out/$TAG/dist_rate.txt
python3 tools/dirty_rate.py 1000 > gpurun_out/$TAG/dirty_rate.txt 2>&1; cat gpurun_out/$TAG/dirty_rate.txt
python3 tools/box_info.py > gpurun_out/$TAG/box_info.txt 2>&1
du -sh gpurun_out/$TAG
