#!/bin/bash
# every BASELINE single-GPU configuration (and the dirty / 10k lines) under rocprofv3: tools/profile_r02.sh each
TAG=${1:-r02_prof}
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_1000x5M -- --steps 20 --warmup 5
bash tools/profile_r02.sh $TAG cfg1_hmh_k16_10000x5M -- --steps 5 --warmup 2 --genomes 10000
bash tools/profile_r02.sh $TAG cfg2_hll_p14_k21_10000x5M -- --steps 5 --warmup 2 --genomes 10000 --algo hll -p 14 -k 21
bash tools/profile_r02.sh $TAG cfg4shape_ull_p12_reads -- --steps 20 --warmup 5 --workload reads --algo ull -p 12 -k 16
bash tools/profile_r02.sh $TAG dirty_nrun_hmh_k16 -- --steps 20 --warmup 5 --dirty nrun
bash tools/profile_r02.sh $TAG dirty_lower_hmh_k16 -- --steps 20 --warmup 5 --dirty lower
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (tools/ubench hbm: 2 GiB copy / read / read64)
OUT=gpurun_out/$TAG/calibration; mkdir -p $OUT; REPO=$(pwd); cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/cal_FETCH -- $REPO/tools/ubench hbm > $REPO/$OUT/cal_FETCH.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/cal_WRITE -- $REPO/tools/ubench hbm > $REPO/$OUT/cal_WRITE.log 2>&1
cd $REPO; python3 tools/pmc_summary.py $OUT > $OUT/calibration.txt 2>&1; rm -rf $OUT/cal_FETCH $OUT/cal_WRITE; cat $OUT/calibration.txt | head -30
$REPO/tools/ubench_hash > gpurun_out/$TAG/ubench_hash.txt 2>&1; cat gpurun_out/$TAG/ubench_hash.txt
du -sh gpurun_out/$TAG
