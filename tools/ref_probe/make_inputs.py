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


if __name__ == "__main__":
    main(sys.argv[1])
