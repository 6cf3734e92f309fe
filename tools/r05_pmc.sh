# PMC passes: the persistent kernel against the sliced one on the same shapes (GPU box)
mkdir -p gpurun_out/r05a
export KERNELS="sketch_kernel"
LASH_SOLE_MAX=1000000 bash tools/pmc_cmd.sh sole_2000x500k tools/one_shape.py 2000 500000 hmh 16 0 > gpurun_out/r05a/pmc_sole_2000x500k.txt 2>&1
LASH_SOLE_MAX=0 bash tools/pmc_cmd.sh reg_2000x500k tools/one_shape.py 2000 500000 hmh 16 0 > gpurun_out/r05a/pmc_reg_2000x500k.txt 2>&1
bash tools/pmc_cmd.sh sole_100kx10k tools/one_shape.py 100000 10000 hmh 16 0 > gpurun_out/r05a/pmc_sole_100kx10k.txt 2>&1
LASH_SOLE_MAX=1000000 bash tools/pmc_cmd.sh sole_hll_2000x500k tools/one_shape.py 2000 500000 hll 21 10 > gpurun_out/r05a/pmc_sole_hll_2000x500k.txt 2>&1
LASH_SOLE_MAX=0 bash tools/pmc_cmd.sh reg_hll_2000x500k tools/one_shape.py 2000 500000 hll 21 10 > gpurun_out/r05a/pmc_reg_hll_2000x500k.txt 2>&1
cat gpurun_out/r05a/pmc_*.txt
