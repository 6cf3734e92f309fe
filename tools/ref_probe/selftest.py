#!/usr/bin/env python3
"""Dry run of the kit WITHOUT the reference: fabricate "reference images" with the oracle under a chosen (non-default) layout,
then check that fit_layout.py recovers exactly that layout.  Proves the search and the report work; proves nothing about lash."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, HERE)


def fabricate(root, spec):
    import numpy as np
    import oracle_lib as O
    from make_inputs import INPUTS
    from fit_layout import load_inputs
    lay = O.parse_layout(spec)
    manifest = {"generated_by": "tools/ref_probe/selftest.py (ORACLE-MADE, not lash)", "inputs": INPUTS, "cases": []}
    seq, off, goff = load_inputs(manifest)
    for line in open(os.path.join(HERE, "cases.tsv")):
        if line.startswith("#") or not line.strip():
            continue
        name, algo, k, p, seed = line.rstrip("\n").split("\t")
        aid = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}[algo]
        img = O.sketch_genomes(aid, int(k), int(p) if algo != "hmh" else 0, int(seed), seq, off, goff, threads=4, layout=lay)
        raw = img.tobytes()
        open(os.path.join(root, name + ".bin"), "wb").write(raw)
        manifest["cases"].append({"name": name, "algo": algo, "k": int(k), "p": int(p), "seed": int(seed), "images": name + ".bin",
                                  "image_bytes": img.shape[1], "sha256": hashlib.sha256(raw).hexdigest()})
    json.dump(manifest, open(os.path.join(root, "manifest.json"), "w"), indent=1)


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "codes=ACTG,kmer=msb,hmh_x=low,hmh_reg=le,hll_bucket=low,hmh_hdr=,hll_hdr=azspl,ull_hdr=pl"
    with tempfile.TemporaryDirectory() as root, tempfile.TemporaryDirectory() as again:
        fabricate(root, spec)
        r = subprocess.run([sys.executable, os.path.join(HERE, "fit_layout.py"), root], capture_output=True, text=True)
        print(r.stdout[-1500:])
        got = open(os.path.join(root, "fitted_layout.txt")).read().strip() if os.path.exists(os.path.join(root, "fitted_layout.txt")) else None
        import oracle_lib as O
        print("wanted:", O.parse_layout(spec).spec(), "\nfitted:", got)
        if got is None:
            return 1
        # (codes, msb) and (complement-reversed codes, lsb) are the same function: compare what the layouts PRODUCE
        fabricate(again, got)
        same = all(open(os.path.join(root, f), "rb").read() == open(os.path.join(again, f), "rb").read()
                   for f in os.listdir(root) if f.endswith(".bin"))
        print("fitted layout reproduces every fabricated image:", same)
        return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
