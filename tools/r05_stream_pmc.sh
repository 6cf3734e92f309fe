#!/bin/bash
# counters of stream_sketch_kernel on soft-masked shapes (GPU box): vector instructions and time per surviving k-mer by block size
TAG=${1:-r05_stream}
LASH_STREAM_FIRST=1 bash tools/profile_py.sh $TAG clean_through_stream tools/dirty_one.py 0 0
bash tools/profile_py.sh $TAG B10000 tools/dirty_one.py 10000 10000
bash tools/profile_py.sh $TAG B2500 tools/dirty_one.py 2500 2500
bash tools/profile_py.sh $TAG B500 tools/dirty_one.py 500 500
for d in clean_through_stream B10000 B2500 B500; do echo "== $d"; cat gpurun_out/$TAG/$d/run.log | tail -1; grep -A12 "pmc_SQ_INSTS_VALU.*stream_sketch" gpurun_out/$TAG/$d/pmc_summary.txt | head -14; grep -A9 "pmc_SQ_ACTIVE_INST_VALU.*stream_sketch" gpurun_out/$TAG/$d/pmc_summary.txt | head -10; grep stream_sketch gpurun_out/$TAG/$d/kernel_stats.csv | cut -c1-140; done
