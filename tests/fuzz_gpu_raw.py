#!/usr/bin/env python3
"""tests/fuzz_gpu_raw.py [iterations] [seed] — randomized parity runs of the device-side FASTA/FASTQ parse
(lash_sketch_files_raw) against a host parse (liblash_host's reader, needletail semantics) + the oracle.
Not collected by pytest; run on the GPU box: `python tests/fuzz_gpu_raw.py 300`."""
import os
import random
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lash_amd
import host_lib as H
import oracle_lib as O
import fuzz_knobs

ALGO = {"hmh": 0, "hll": 1, "ull": 2}


def seq_bytes(rng, n):
    s = bytearray(O.synth_genome(rng.randint(0, 10**6), max(n, 1)).tobytes()[:n])
    mode = rng.random()
    if mode < 0.3 and n:
        for _ in range(rng.randint(1, 4)):
            i = rng.randrange(n)
            s[i:i + rng.choice([1, 3, 50])] = b"N" * len(s[i:i + rng.choice([1, 3, 50])])
    elif mode < 0.45:
        s = bytearray(bytes(s).lower()) if rng.random() < 0.3 else bytearray(b"".join(bytes([c | 0x20]) if rng.random() < 0.3 else bytes([c]) for c in s))
    return bytes(s)


def fasta_file(rng):
    nl = rng.choice([b"\n", b"\n", b"\r\n"])
    out = []                                                # (the first byte of a file must be '>' or '@': needletail)
    for r in range(rng.randint(0, 12)):
        hdr = b">" + rng.choice([b"r%d" % r, b"r%d ACGT ACGT description with > inside" % r, b"", b"x" * rng.choice([5, 300, 20000])])
        out.append(hdr + nl)
        s = seq_bytes(rng, rng.choice([0, rng.randint(1, 300), rng.randint(1, 60000)]))
        w = rng.choice([60, 70, 80, 1, 17, 10**9])
        if s and rng.random() < 0.15:                       # a '>' (or '@') in the middle of a sequence line is sequence text
            i = rng.randrange(len(s))
            s = s[:i] + rng.choice([b">", b">r9 x", b"@"]) + s[i:]
        for i in range(0, len(s), w):
            out.append(s[i:i + w] + nl)
            if rng.random() < 0.02:
                out.append(nl)                              # blank line inside a record
    data = b"".join(out)
    if rng.random() < 0.3 and data.endswith(nl):
        data = data[:-len(nl)]                              # no trailing newline
    return data


def fastq_file(rng, length_errors=True):
    nl = rng.choice([b"\n", b"\n", b"\r\n"])
    out = []
    for r in range(rng.randint(0, 400)):
        s = seq_bytes(rng, rng.choice([0, 36, 100, 150, 151, rng.randint(1, 400)]))
        q = bytes(rng.choice(b"@+>IIIIFF#ACGT") for _ in range(len(s)))
        out.append(b"@" + rng.choice([b"r%d" % r, b"r%d/1 @+ACGT" % r]) + nl + s + nl + b"+" + rng.choice([b"", b"r%d" % r]) + nl + q + nl)
    if out and rng.random() < 0.25:                         # malformed: needletail stops there, the records before it stand
        i = rng.randrange(len(out))
        kind = rng.random()
        if kind < 0.3:
            out.insert(i, nl)                               # blank line between records
        elif kind < 0.6:
            out[i] = out[i].replace(nl + b"+", nl + b"-", 1)   # the '+' line does not start with '+'
        elif kind < 0.8 or not length_errors:
            out[i] = b"r" + out[i][1:]                      # header without '@'
        else:                                               # quality one character short: needletail's other FASTQ error; the
            out[i] = out[i][:-len(nl) - 1] + nl if len(out[i]) > len(nl) + 1 else out[i]   # 4-line structure stays intact (fastq_check.hip)
    data = b"".join(out)
    if rng.random() < 0.3 and data.endswith(nl) and out and len(s):
        data = data[:-len(nl)]                              # no newline after the last quality line (if it is not empty:
    return data                                             # an empty last line would vanish and truncate the record)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = lash_amd.Context(0)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        for it in range(iters):
            rng = random.Random(seed0 * 7919 + it)
            knobs = fuzz_knobs.set_sole(random.Random(seed0 * 1000003 + it))    # which genomes go to the persistent small-genome kernel (FUZZ_SOLE)
            an = rng.choice(["hmh", "hll", "ull"])
            if os.environ.get("FUZZ_ALGO"):
                an = os.environ["FUZZ_ALGO"]
            k = rng.choice([rng.randint(1, 32), 16, 21])
            p = 0 if an == "hmh" else rng.randint(4, 14)
            if os.environ.get("FUZZ_P") and an != "hmh":             # pin the precision (register tables beyond LDS: tools/bins_fuzz.sh)
                p = min(rng.choice([int(x) for x in os.environ["FUZZ_P"].split(",")]), 16 if an == "hll" else 26)
            files = [fasta_file(rng) if rng.random() < 0.6 else fastq_file(rng) for _ in range(rng.randint(1, 6))]
            files = [f for f in files if f[:1] in (b">", b"@")] or [b">only\nACGT\n"]
            got = ctx.sketch_files_raw(an, k, p, 42, files)
            want = O.sketch_files(ALGO[an], k, p, 42, files, threads=8)   # the oracle's own needletail-like parse (stops at a malformed record)
            gs = []
            for i, f in enumerate(files):
                path = os.path.join(td, "f%d" % i)
                open(path, "wb").write(f)
                gs.append(H.read_fastx(path))               # the C++ host reader must agree with it
            seq, off, goff = lash_amd.records_to_arrays(gs)
            assert np.array_equal(want, O.sketch_genomes(ALGO[an], k, p, 42, seq, off, goff, threads=8)), "host reader != oracle parse (it=%d)" % it
            if not np.array_equal(got, want):
                bad = sorted({int(r) for r in np.argwhere(got != want)[:, 0]})
                for b in bad:
                    open("/tmp/fuzz_raw_fail_%d.bin" % b, "wb").write(files[b])
                print("MISMATCH [" + knobs + "] it=%d %s k=%d p=%d files=%s (saved to /tmp/fuzz_raw_fail_*.bin)" % (it, an, k, p, bad))
                sys.exit(1)
    print("raw fuzz ok: %d iterations from seed %d" % (iters, seed0))


if __name__ == "__main__":
    main()
