OUT=gpurun_out/r05a; mkdir -p $OUT
export KERNELS="stream_sketch"
bash tools/pmc_cmd.sh dirty_lower bench.py --steps 10 --warmup 3 --genomes 1000 --dirty lower --no-cpu-baseline --no-ubench --no-parity-check 2>&1 | tee $OUT/pmc_dirty_lower.txt
LASH_DEFER_MIN=-1 bash tools/pmc_cmd.sh dirty_lower_nodefer bench.py --steps 10 --warmup 3 --genomes 1000 --dirty lower --no-cpu-baseline --no-ubench --no-parity-check 2>&1 | tee $OUT/pmc_dirty_lower_nodefer.txt
