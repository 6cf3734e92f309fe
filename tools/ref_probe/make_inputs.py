#!/usr/bin/env python3
"""Inputs of the probe: the committed fixtures plus the SURVEY Appendix-B sequence as a one-record FASTA."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")
INPUTS = ["appendix_b.fasta", "fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq"]
# the map-order probe: enough names for the hash map to grow 4 -> 8 -> 16 -> 32 -> 64 buckets, two of them long enough
# for XXH3's 129-240 and > 240 byte classes
ORDER_NAMES = ["o%02d.fasta" % i for i in range(38)] + ["o_" + "x" * 150 + ".fasta", "o_" + "y" * 245 + ".fasta"]
# the error probe: lash's record loop keeps calling next() after an Err (utils.rs:457-458), so what it sees after
# a malformed FASTQ record is needletail's business.  Two files with good records on BOTH sides of one malformed record; every
# record has its own sequence, so the image tells whether the records after the error were sketched.
ERROR_INPUTS = ["err_bad_plus.fastq", "err_short_quality.fastq"]


def error_files():
    """name -> bytes.  Deterministic: sequences from a small LCG, 60 + 7 i bases each."""
    def seq(i):
        x, out = 12345 + 977 * i, bytearray()
        for _ in range(60 + 7 * i):
            x = (x * 1103515245 + 12345) & 0x7FFFFFFF
            out.append(b"ACGT"[(x >> 16) & 3])
        return bytes(out)

    def rec(i, plus=b"+", qual=None):
        s = seq(i)
        return b"@e%d\n%s\n%s\n%s\n" % (i, s, plus, (b"I" * len(s)) if qual is None else qual)
    return {"err_bad_plus.fastq": rec(0) + rec(1) + rec(2) + rec(3, plus=b"-") + rec(4) + rec(5) + rec(6),
            "err_short_quality.fastq": rec(10) + rec(11) + rec(12, qual=b"IIII") + rec(13) + rec(14) + rec(15)}


def main(workdir):
    os.makedirs(workdir, exist_ok=True)
    with open(os.path.join(workdir, "appendix_b.fasta"), "w") as f:
        f.write(">appendix_b\nACGTTGCATGCATCGATCGGATTACA\n")
    for name in INPUTS[1:]:
        with open(os.path.join(GOLD, name), "rb") as src, open(os.path.join(workdir, name), "wb") as dst:
            dst.write(src.read())
    with open(os.path.join(workdir, "list.txt"), "w") as f:          # relative names: lash dist prints them, the tests compare them
        f.write("\n".join(INPUTS) + "\n")
    order = workdir.rstrip("/") + "_order"
    os.makedirs(order, exist_ok=True)
    for i, name in enumerate(ORDER_NAMES):
        src = os.path.join(GOLD, INPUTS[1 + i % 3])
        with open(src, "rb") as a, open(os.path.join(order, name), "wb") as b:
            b.write(a.read())
    with open(os.path.join(order, "list.txt"), "w") as f:
        f.write("\n".join(ORDER_NAMES) + "\n")
    errors = workdir.rstrip("/") + "_errors"
    os.makedirs(errors, exist_ok=True)
    for name, data in error_files().items():
        with open(os.path.join(errors, name), "wb") as f:
            f.write(data)
    with open(os.path.join(errors, "list.txt"), "w") as f:
        f.write("\n".join(ERROR_INPUTS) + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
