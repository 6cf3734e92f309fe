# tools/sole_threads_ab.sh — the persistent kernel's workgroup shapes against each other (GPU box): LASH_SOLE_THREADS x genome size
OUT=gpurun_out/r05a; mkdir -p $OUT
for T in 64 128 256 512; do
  echo "== LASH_SOLE_THREADS=$T"
  LASH_SOLE_THREADS=$T timeout 600 python3 tools/small_genomes_rate.py 2>&1 | grep -v amdgpu.ids
done
