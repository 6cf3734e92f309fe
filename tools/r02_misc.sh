#!/bin/bash
# full GPU suite, async host-entry rate (the sanitizer stress runs on the CPU build only: this pool refuses sanitizer builds)
OUT=gpurun_out/${1:-r02_misc}; mkdir -p $OUT
timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
python3 tools/host_entry_rate.py > $OUT/host_entry_rate.txt 2>&1; cat $OUT/host_entry_rate.txt
