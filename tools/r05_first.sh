OUT=gpurun_out/r05a; mkdir -p $OUT
bash tools/ktrace_py.sh tools/viral_rate.py 2>&1 | tee $OUT/viral_ktrace.txt
export KERNELS="sole_sketch"
bash tools/pmc_cmd.sh viral tools/viral_rate.py 2>&1 | tee $OUT/viral_pmc.txt
