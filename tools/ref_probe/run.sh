#!/bin/bash
# tools/ref_probe/run.sh — produce reference images with the REAL `lash` and fit the layout switches to them.
# Needs: cargo with a nightly toolchain (or LASH_BIN=/path/to/lash), python3, and `zstd` or python's `zstandard`.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
WORK=${WORK:-$HERE/_work}
OUT=$REPO/tests/golden/ref_images
mkdir -p "$WORK" "$OUT"

if [ -z "${LASH_BIN:-}" ]; then
    command -v cargo >/dev/null || { echo "no cargo and no LASH_BIN: cannot build the reference here" >&2; exit 3; }
    # the exact dependency set of /root/reference/Cargo.lock (lash-rs 0.1.6 on crates.io ships that lock file)
    cargo +nightly install lash-rs --version 0.1.6 --locked --root "$HERE/_build"
    LASH_BIN=$HERE/_build/bin/lash
fi
"$LASH_BIN" --version | tail -1 > "$WORK/lash_version.txt" || true

# HLL++ bias tables: streaming_algorithms' source is unpacked in the cargo registry once lash-rs has been built; without
# them `lash dist` of this repository refuses HyperLogLog estimates <= 5 * 2^p (the hll_p14_k21 case below is in that regime)
for reg in "${CARGO_HOME:-$HOME/.cargo}"/registry/src/*/streaming_algorithms-0.3.3 ${STREAMING_ALGORITHMS_SRC:-}; do
    [ -d "$reg" ] || continue
    python3 "$HERE/extract_hll_bias.py" "$reg" "$OUT/hllpp_bias.txt" && break || echo "no bias tables found in $reg" >&2
done
[ -f "$OUT/hllpp_bias.txt" ] || echo "HLL++ bias tables not extracted (set STREAMING_ALGORITHMS_SRC=<crate dir>): hll small-range rows stay refused" >&2

python3 "$HERE/make_inputs.py" "$WORK/in"
grep -v '^#' "$HERE/cases.tsv" | while IFS=$'\t' read -r name algo k p seed; do
    [ -n "$name" ] || continue
    d=$WORK/$name; rm -rf "$d"; mkdir -p "$d"
    cp "$WORK"/in/* "$d"/
    ( cd "$d"
      "$LASH_BIN" sketch -f list.txt -o "$name" -a "$algo" -k "$k" -p "$p" -s "$seed" -t 2 > sketch.log 2>&1
      # dist discovers its three input files by prefix in the current directory (main.rs:284-337)
      "$LASH_BIN" dist -q "$name" -r "$name" -o "$name.dist.raw" -t 2 > dist.log 2>&1 || echo "dist failed for $name" >&2
      if [ "$algo" = ull ]; then "$LASH_BIN" dist -q "$name" -r "$name" -e ml -o "$name.dist_ml.raw" -t 2 >> dist.log 2>&1 || true; fi
    )
done
# map-order probe: 40 names; `dist --dm` prints its columns in the query map's key order, and with one rayon thread its
# rows in the reference map's (utils.rs:111-160) — pins lash_amd/csrc/host/name_order.cpp
d=$WORK/order; rm -rf "$d"; mkdir -p "$d"; cp "$WORK"/in_order/* "$d"/
( cd "$d"
  "$LASH_BIN" sketch -f list.txt -o order -a hmh -k 16 -t 2 > sketch.log 2>&1
  "$LASH_BIN" dist -q order -r order --dm -t 1 -o order.dm.raw > dist.log 2>&1
  "$LASH_BIN" dist -q order -r order -t 1 -o order.rows.raw >> dist.log 2>&1
) || echo "map-order probe failed" >&2
# error probe: what does lash sketch AFTER a malformed FASTQ record? (utils.rs:457-458 keeps calling next(); pins layout.fastq_skip_bad)
d=$WORK/errors; rm -rf "$d"; mkdir -p "$d"; cp "$WORK"/in_errors/* "$d"/
( cd "$d"
  "$LASH_BIN" sketch -f list.txt -o errors -a hmh -k 16 -t 1 > sketch.log 2>&1
) || echo "error probe failed (a panic on malformed FASTQ would itself be the answer: see $d/sketch.log)" >&2
python3 "$HERE/collect.py" "$WORK" "$OUT"
( cd "$HERE/kmer_probe" && cargo +nightly run --release > "$OUT/kmer_probe.txt" 2> "$WORK/kmer_probe.err" ) || echo "kmer_probe did not build/run (optional)" >&2
python3 "$HERE/fit_layout.py" "$OUT"
