#!/usr/bin/env python3
"""Inputs of the probe: the committed fixtures plus the SURVEY Appendix-B sequence as a one-record FASTA."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")
INPUTS = ["appendix_b.fasta", "fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq"]


def main(workdir):
    os.makedirs(workdir, exist_ok=True)
    with open(os.path.join(workdir, "appendix_b.fasta"), "w") as f:
        f.write(">appendix_b\nACGTTGCATGCATCGATCGGATTACA\n")
    for name in INPUTS[1:]:
        with open(os.path.join(GOLD, name), "rb") as src, open(os.path.join(workdir, name), "wb") as dst:
            dst.write(src.read())
    with open(os.path.join(workdir, "list.txt"), "w") as f:          # relative names: lash dist prints them, the tests compare them
        f.write("\n".join(INPUTS) + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
