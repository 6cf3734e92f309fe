#!/usr/bin/env python3
"""Write the cfg1 plumbing fixtures (BASELINE.json configs[0]) and their oracle sketches' digests.

The reference mentions ./data/*.fasta (README.md:103) but ships no data directory, so the
fixtures are generated here: three ~50 kbp FASTA files with N runs, IUPAC codes, lower-case
(soft-masked) stretches, a record shorter than k, an empty record and CRLF line ends.
Expected digests come from the CPU oracle (oracle/lash_oracle.c) and pin GPU == oracle == pyref.
"""
import hashlib, json, os, random, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import oracle_lib as O
from fastx import read_fastx


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def wrap(s, w=80, eol="\n"):
    return eol.join(s[i:i + w] for i in range(0, len(s), w)) + eol


def main():
    rng = random.Random(20260128)
    # fixture A: 2 records, N runs + IUPAC + a lower-case block
    a1 = list(rand_seq(rng, 30000))
    for pos, ln in [(100, 1), (5000, 37), (12000, 500), (29990, 10)]:
        a1[pos:pos + ln] = "N" * ln
    for pos, ch in [(7, "R"), (800, "Y"), (801, "K"), (20000, "n")]:
        a1[pos] = ch
    a1[15000:15400] = [c.lower() for c in a1[15000:15400]]
    a2 = rand_seq(rng, 20000)
    fa = ">A1 with N runs\n" + wrap("".join(a1)) + ">A2 clean\n" + wrap(a2, 60)
    # fixture B: many short records, one shorter than k (10 bp), one empty, one all-N
    recs = [rand_seq(rng, rng.randint(16, 400)) for _ in range(200)]
    recs.insert(3, "ACGTACGTAC")
    recs.insert(10, "")
    recs.insert(20, "N" * 50)
    recs.insert(30, rand_seq(rng, 15) + "N" + rand_seq(rng, 15))   # 30 valid bases joined across the N
    recs.append(rand_seq(rng, 12000))
    fb = "".join(">r%d\n%s" % (i, wrap(r, 70) if r else "\n") for i, r in enumerate(recs))
    # fixture C: single 50 kbp record, CRLF line ends, trailing lower-case
    c = rand_seq(rng, 50000)
    c = c[:49000] + c[49000:].lower()
    fc = ">C single\r\n" + wrap(c, 80, "\r\n")
    names = ["fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta"]
    for n, txt in zip(names, [fa, fb, fc]):
        with open(os.path.join(HERE, n), "w", newline="") as f:
            f.write(txt)
    # FASTQ twin of fixture B's first 40 records (needletail auto-detects '@')
    fq = "".join("@q%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(recs[:40]))
    with open(os.path.join(HERE, "fixture_B40.fastq"), "w") as f:
        f.write(fq)
    names.append("fixture_B40.fastq")

    digests = {}
    cases = [("hmh", O.HMH, 16, 0), ("hmh", O.HMH, 21, 0), ("hmh", O.HMH, 11, 0), ("hll", O.HLL, 21, 14),
             ("hll", O.HLL, 16, 10), ("ull", O.ULL, 16, 12), ("ull", O.ULL, 31, 10), ("ull", O.ULL, 32, 8)]
    for n in names:
        recs_b = read_fastx(os.path.join(HERE, n))
        seq = np.frombuffer(b"".join(recs_b), dtype=np.uint8)
        off = np.cumsum([0] + [len(r) for r in recs_b]).astype(np.uint64)
        for an, algo, k, p in cases:
            img = O.sketch_genomes(algo, k, p, 42, seq, off, np.array([0, len(recs_b)], np.uint64))[0]
            digests["%s|%s|k%d|p%d|seed42" % (n, an, k, p)] = hashlib.sha256(img.tobytes()).hexdigest()
    with open(os.path.join(HERE, "fixture_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    print("wrote", len(digests), "digests")


if __name__ == "__main__":
    main()
