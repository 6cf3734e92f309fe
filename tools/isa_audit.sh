#!/bin/bash
# tools/isa_audit.sh [out_dir] — the issue-cost audit of the three judged sketch kernels: compile
# sketch_kernels.hip to a listing, price the hot blocks of each kernel with the measured cost table
# (profiles/r04/isa_cost/costs.json <- tools/ubench_isa on the GPU) and write one text + one JSON file per kernel.
# The measured figures beside the predictions come from the rocprofv3 passes under profiles/r05/ (cycles = kernel time x clock
# / wave-k-mers per SIMD; SQ_INSTS_VALU / k-mers x 64).  Runs on the build machine (no GPU needed).
# Round 5: the listing is compiled with -gline-tables-only (same code, .loc lines added) and every section is
# printed with its MNEMONIC histogram and the source lines its VALU instructions were written on (--mnemonics); the HyperLogLog
# selector now picks the UNMASKED k = 21 body (round 4's picked the masked one: profiles/r05/floor_hll_ull.md).
set -e
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r05/isa_cost}; mkdir -p "$OUT"
S=${LISTING:-/tmp/lash_sketch_kernels_loc.s}
[ -n "$LISTING" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -gline-tables-only -S --cuda-device-only -o "$S" lash_amd/csrc/sketch_kernels.hip 2>/dev/null
# measured figures from the round's counter passes when they are there ("--measured C --measured-valu V": tools/isa_measured.py)
P=${PROFILES:-profiles/r05}
[ -n "$M_HMH" ] || M_HMH=$(python3 tools/isa_measured.py $P/default_hmh_k16_12500x5M/pmc_summary.txt "lash::sketch_kernel<0, 0, false, 0, true" $((12500*4999985)) 2>/dev/null || true)
[ -n "$M_HLL" ] || M_HLL=$(python3 tools/isa_measured.py $P/cfg2_hll_p14_k21_10000x5M/pmc_summary.txt "lash::sketch_kernel<1, 2" $((10000*4999980)) 2>/dev/null || true)
[ -n "$M_ULL" ] || M_ULL=$(python3 tools/isa_measured.py $P/cfg4shape_ull_p12_reads/pmc_summary.txt "lash::sketch_kernel<2, 0" $((20000000*135)) 2>/dev/null || true)
MF="--mnemonics --mix-factor profiles/r04/isa_cost/mix_factor.json"
# HyperMinHash k = 16, direct, deferring (bench.py's default): the filter's four groups of four k-mers, the drain round
# (0.028 k-mers pass / 0.64 lanes busy per round = 0.044 rounds per k-mer), the tile's ASCII -> 2-bit conversion
python3 tools/isa_cost.py "$S" --kernel 'sketch_kernel<0, 0, false, 0, true, true>' $M_HMH $MF --json "$OUT/hmh_k16_defer.json" \
  --section 'filter: window, rank half of xxh3_128, threshold test, append (unmasked)|16|mul==28&bfe_i32==0&ds_write_b32==4' \
  --section 'drain round: pop + full xxh3_128 + threshold word + ds_min (0.044 rounds per k-mer)|22.7|mul==16&ds_min_u32>=1@first' \
  --section 'tile: ASCII -> 2-bit words (64 k-mers per lane)|64|perm>=40@first' > "$OUT/hmh_k16_defer.txt"
# HyperLogLog p = 14, k = 21 (BASELINE configs[2]): the k = 21 word body, the tile's conversion (six chunks)
python3 tools/isa_cost.py "$S" --kernel 'sketch_kernel<1, 2, false, 0, true, false>' $M_HLL $MF --json "$OUT/hll_p14_k21.json" \
  --section 'word: 16 x (64-bit window k = 21, xxh3_64, rank, ds_max) (unmasked)|16|mul==96&bfe==0&alignbit==107' \
  --section '~word: the same, masked (tiles with a record boundary or a non-ACGT byte; round 4 priced this one)|16|mul==96&bfe_i32==0&bfe>=16' \
  --section 'tile: ASCII -> 2-bit words (64 k-mers per lane)|64|perm>=48@first' > "$OUT/hll_p14_k21.txt"
# UltraLogLog p = 12, k = 16 on reads (configs[4] shape): every tile holds record boundaries -> the masked fast body; of a 150-bp read's
# 150 window starts 135 are k-mers, and the census counts k-mers: 16 starts = 14.4 k-mers
python3 tools/isa_cost.py "$S" --kernel 'sketch_kernel<2, 0, false, 0, true, false>' $M_ULL $MF --json "$OUT/ull_p12_k16_reads.json" \
  --section 'word: 16 starts = 14.4 k-mers x (window, xxh3_64, nlz, ds_or) (masked fast form)|14.4|mul==96&bfe_i32==16&ffbh==16' \
  --section 'tile: ASCII -> 2-bit words (64 starts = 57.6 k-mers per lane)|57.6|perm>=40@first' > "$OUT/ull_p12_k16_reads.txt"
for f in "$OUT"/*.txt; do echo "== $f"; tail -n 3 "$f"; done
