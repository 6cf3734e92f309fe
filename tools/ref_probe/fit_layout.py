#!/usr/bin/env python3
"""Fit `lash_layout` to images written by the real `lash`.

usage: fit_layout.py [tests/golden/ref_images]

For every case of the manifest the CPU oracle sketches the same inputs under every candidate layout and the result is
compared with the reference bytes, in two stages so that the search stays small and the report says WHICH rule is off:
  1. register area (the last 32 768 / 2^p bytes of each image): base codes (24 permutations) x k-mer bit order (2) x
     HyperMinHash xxh3 half and register byte order (2 x 2) or HyperLogLog bucket side (2);
  2. header (what precedes the registers): depth-first over the field codes of `lash_layout`, each field accepted only
     if its encoding of the known value equals the reference bytes at that offset.
Prints the fitted layout per sketch type, the combined spec string, and how it differs from the repository default.
Exit status 0 = every case explained by one layout, 1 = not.
"""
import itertools
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))


def load_inputs(manifest):
    from fastx import read_fastx
    import numpy as np
    gold = os.path.join(REPO, "tests", "golden")
    recs_per_file = []
    for name in manifest["inputs"]:
        path = os.path.join(gold, name)
        if name == "appendix_b.fasta" and not os.path.exists(path):
            recs_per_file.append([b"ACGTTGCATGCATCGATCGGATTACA"])
        else:
            recs_per_file.append(read_fastx(path))
    flat = [r for recs in recs_per_file for r in recs]
    seq = np.frombuffer(b"".join(flat), dtype=np.uint8)
    off = np.cumsum([0] + [len(r) for r in flat]).astype(np.uint64)
    goff = np.cumsum([0] + [len(recs) for recs in recs_per_file]).astype(np.uint64)
    return seq, off, goff


def header_candidates(ref_hdr, values):
    """all templates (strings of field codes) whose encoding equals ref_hdr; values: code -> bytes"""
    found = []

    def walk(at, tpl):
        if at == len(ref_hdr):
            found.append(tpl)
            return
        if len(tpl) >= 7:
            return
        for code, enc in values.items():
            if ref_hdr[at:at + len(enc)] == enc:
                walk(at + len(enc), tpl + code)
    walk(0, "")
    return found


def main():
    import numpy as np
    import oracle_lib as O
    root = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden", "ref_images")
    manifest = json.load(open(os.path.join(root, "manifest.json")))
    seq, off, goff = load_inputs(manifest)
    n_files = len(goff) - 1
    algo_id = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}
    default = O.default_layout()
    fitted = {}          # algo -> set of (codes, kmer, algo-specific switches, header template)
    ok_all = True
    for case in manifest["cases"]:
        an, k, p, seed = case["algo"], case["k"], case["p"], case["seed"]
        ref = np.frombuffer(open(os.path.join(root, case["images"]), "rb").read(), dtype=np.uint8).reshape(n_files, -1)
        n_reg_bytes = 32768 if an == "hmh" else 1 << p
        hdr_len = ref.shape[1] - n_reg_bytes
        print("== %s: %d files x %d bytes (header %d + registers %d)" % (case["name"], n_files, ref.shape[1], hdr_len, n_reg_bytes))
        if hdr_len < 0:
            print("   image smaller than its register array: the register width / count hypothesis is wrong")
            ok_all = False
            continue
        ref_regs = ref[:, hdr_len:]
        matches, best = [], None
        for perm in itertools.permutations("ACGT"):
            codes = "".join(perm)
            for kmer in ("msb", "lsb"):
                extra = ([dict(hmh_x=x, hmh_reg=e) for x in ("high", "low") for e in ("le", "be")] if an == "hmh" else
                         [dict(hll_bucket=b) for b in ("low", "high")] if an == "hll" else [dict()])
                for ex in extra:
                    lay = O.make_layout(codes=codes, kmer=kmer, hmh_hdr="", hll_hdr="", ull_hdr="", **ex)
                    img = O.sketch_genomes(algo_id[an], k, p if an != "hmh" else 0, seed, seq, off, goff, threads=4, layout=lay)
                    diff = int((img != ref_regs).sum())
                    key = (codes, kmer, tuple(sorted(ex.items())))
                    if diff == 0:
                        matches.append(key)
                    if best is None or diff < best[0]:
                        occ = float(((img != 0) == (ref_regs != 0)).mean())
                        best = (diff, key, occ)
        if not matches:
            ok_all = False
            diff, key, occ = best
            print("   NO register rule in the search space reproduces the reference registers.")
            print("   closest: %s  -> %d of %d register bytes differ; zero/non-zero pattern agrees on %.2f %% of bytes"
                  % (key, diff, ref_regs.size, 100 * occ))
            print("   reading: pattern agreement near 100 %% with different values implicates the rank / signature rule "
                  "(lz, rho, nlz, sig); a different pattern implicates the k-mer value, the hashed bytes or the bucket rule.")
            continue
        # (codes, msb) and (complement-reversed codes, lsb) describe the same canonical k-mers: two candidates are normal.
        # Prefer the one closest to the repository default.
        dflt = {"hmh_x": "high", "hmh_reg": "le", "hll_bucket": "low"}
        matches.sort(key=lambda m: (m[1] != "msb") + (m[0] != "ACGT") + sum(v != dflt[kk] for kk, v in m[2]))
        print("   registers reproduced by %d candidate(s): %s" % (len(matches), matches[:6]))
        # ---- header ----
        codes, kmer, ex = matches[0]
        lay = O.make_layout(codes=codes, kmer=kmer, hmh_hdr="", hll_hdr="", ull_hdr="", **dict(ex))
        tpls = None
        for f in range(n_files):
            regs = ref_regs[f]
            pp = 14 if an == "hmh" else p
            n = 16384 if an == "hmh" else 1 << p
            alpha = {4: 0.673, 5: 0.697, 6: 0.709}.get(pp, 0.7213 / (1.0 + 1.079 / (1 << pp)))
            zero = int((regs == 0).sum()) if an == "hll" else 0
            # the same summation order as the oracle's save: exact powers of two, any order gives the same f64 here
            tot = float(np.sum(2.0 ** -regs.astype(np.float64))) if an == "hll" else 0.0
            values = {"p": struct.pack("<B", pp), "P": struct.pack("<I", pp), "Q": struct.pack("<Q", pp),
                      "l": struct.pack("<Q", n), "L": struct.pack("<I", n)}
            if an == "hll":
                values.update({"a": struct.pack("<d", alpha), "z": struct.pack("<Q", zero), "Z": struct.pack("<I", zero),
                               "s": struct.pack("<d", tot)})
            cand = set(header_candidates(bytes(ref[f, :hdr_len]), values))
            tpls = cand if tpls is None else tpls & cand
        if not tpls:
            ok_all = False
            print("   header (%d bytes) is not a sequence of known fields; first file's header bytes: %s"
                  % (hdr_len, bytes(ref[0, :hdr_len]).hex()))
            continue
        tpl = sorted(tpls, key=len)[0]
        print("   header template(s): %s -> using %r" % (sorted(tpls), tpl))
        fitted.setdefault(an, set()).add((codes, kmer, ex, tpl))
    print()
    if not ok_all or any(len(v) != 1 for v in fitted.values()):
        print("RESULT: not every case is explained by one layout (see above).")
        for an, v in fitted.items():
            print("  %s candidates: %s" % (an, sorted(v)))
        return 1
    kw = {}
    for an, v in fitted.items():
        codes, kmer, ex, tpl = next(iter(v))
        if kw.get("codes", codes) != codes or kw.get("kmer", kmer) != kmer:
            print("RESULT: sketch types disagree on the k-mer rule: %s vs %s" % ((kw["codes"], kw["kmer"]), (codes, kmer)))
            return 1
        kw.update(codes=codes, kmer=kmer, **dict(ex))
        kw[an + "_hdr"] = tpl
    lay = O.make_layout(**kw)
    print("RESULT: one layout reproduces every reference image:")
    print("   " + lay.spec())
    d = dict(item.split("=") for item in default.spec().split(","))
    g = dict(item.split("=") for item in lay.spec().split(","))
    changed = {key: (d[key], g[key]) for key in d if d[key] != g[key]}
    if not changed:
        print("   == the repository default: the oracle and the HIP path are now pinned to `lash` on these inputs.")
    else:
        names = {"codes": "U5 base codes", "kmer": "U5 k-mer bit order", "hmh_x": "U1 xxh3_128 half", "hmh_reg": "U2 register byte order",
                 "hll_bucket": "U3 bucket side", "hmh_hdr": "U2 header", "hll_hdr": "U3 header", "ull_hdr": "U4 header"}
        for key, (was, now) in changed.items():
            print("   SWITCH %-10s (%s): default %r -> reference %r" % (key, names[key], was, now))
        print("   set kDefaultLayout (lash_amd/csrc/lash_api.hip) and DEFAULT_LAYOUT (oracle/lash_oracle.c) accordingly; until then run "
              "with --layout '%s'" % lay.spec())
    with open(os.path.join(root, "fitted_layout.txt"), "w") as f:
        f.write(lay.spec() + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
