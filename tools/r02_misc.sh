#!/bin/bash
# full GPU suite, async host-entry rate, TSan stress with real contexts
OUT=gpurun_out/${1:-r02_misc}; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
python3 tools/host_entry_rate.py > $OUT/host_entry_rate.txt 2>&1; cat $OUT/host_entry_rate.txt
TSAN_DIR=/tmp/lash_tsan timeout 900 bash tools/tsan_multictx.sh 4 12 > $OUT/tsan.txt 2>&1; tail -3 $OUT/tsan.txt
cp /tmp/lash_tsan/run.err $OUT/tsan_run.err 2>/dev/null; grep -c "WARNING: ThreadSanitizer" $OUT/tsan_run.err
