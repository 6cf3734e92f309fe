#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the hll k=21 direct kernel (GPU box)
REPO=$(pwd); OUT=$REPO/gpurun_out/wr_hll; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for PMC in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_$PMC -- python3 $REPO/bench.py --genomes 2000 --algo hll -k 21 -p 14 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check --no-ubench > $OUT/$PMC.log 2>&1
done
cd $REPO; python3 tools/pmc_summary.py $OUT "sketch_kernel<1, 2, false, 0, true" | grep -E "dispatches|SIZE"
