#!/usr/bin/env python3
"""tools/make_reads_gz.py OUT.fastq.gz GBP [MEMBERS] [PROCS] — synthetic metagenome-shaped input of BASELINE configs[4]:
150-bp reads of uniform random ACGT (1 N in 5 000 bases, as sequencers emit), constant quality 'I', written as MEMBERS
concatenated gzip members (level 1) — a valid .gz file for any reader.  Prints reads, text bytes and file bytes."""
import os
import sys
import zlib
from multiprocessing import Pool

import numpy as np

RL = 150


def member_bases(idx, n_reads):
    """The reads of member idx as one uint8 array of n_reads * RL bases (deterministic in idx)."""
    rng = np.random.default_rng(20260128 + idx)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=n_reads * RL, dtype=np.uint8)]
    bases[rng.integers(0, n_reads * RL, size=n_reads * RL // 5000)] = ord("N")
    return bases


def member(args):
    idx, n_reads = args
    bases = member_bases(idx, n_reads)
    hdr = 15                                                     # "@m0000r0000000\n"
    rec = hdr + (RL + 1) + 2 + (RL + 1)
    out = np.empty((n_reads, rec), np.uint8)
    ids = ("@m%04dr%07d\n" % (idx % 10000, 0)).encode()
    out[:, :hdr] = np.frombuffer(ids, np.uint8)
    digits = np.arange(n_reads)
    for d in range(7):                                           # the 7-digit read number
        out[:, hdr - 2 - d] = 48 + (digits // 10 ** d) % 10
    out[:, hdr:hdr + RL] = bases.reshape(n_reads, RL)
    out[:, hdr + RL] = 10
    out[:, hdr + RL + 1] = ord("+")
    out[:, hdr + RL + 2] = 10
    out[:, hdr + RL + 3:hdr + 2 * RL + 3] = ord("I")
    out[:, rec - 1] = 10
    co = zlib.compressobj(1, zlib.DEFLATED, 31)                  # 31: gzip container
    return co.compress(out.tobytes()) + co.flush(), n_reads * rec


def main():
    path, gbp = sys.argv[1], float(sys.argv[2])
    members = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else min(16, os.cpu_count() or 1)
    n_reads = int(gbp * 1e9) // RL
    per = (n_reads + members - 1) // members
    jobs = [(i, min(per, n_reads - i * per)) for i in range(members) if n_reads - i * per > 0]
    text = 0
    with Pool(procs) as pool, open(path, "wb") as f:
        for blob, tb in pool.imap(member, jobs):
            f.write(blob)
            text += tb
    print("reads %d text_bytes %d file_bytes %d members %d" % (n_reads, text, os.path.getsize(path), len(jobs)))


if __name__ == "__main__":
    main()
