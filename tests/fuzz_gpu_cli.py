#!/usr/bin/env python3
"""tests/fuzz_gpu_cli.py [iterations] [seed] — `lash sketch` on random file sets (FASTA / FASTQ, plain / gzip / bzip2 / xz /
zstd, many small batches, one or several GPU workers) against a host parse + the oracle.  GPU box, manual."""
import bz2
import gzip
import lzma
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lash_amd
import host_lib as H
import oracle_lib as O
import fuzz_knobs
from fuzz_gpu_raw import fasta_file, fastq_file

ALGO = {"hmh": 0, "hll": 1, "ull": 2}


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        for it in range(iters):
            rng = random.Random(seed0 * 15485863 + it)
            knobs = fuzz_knobs.set_sole(random.Random(seed0 * 1000003 + it))    # which genomes go to the persistent small-genome kernel (FUZZ_SOLE)
            paths, recs = [], []
            for i in range(rng.randint(1, 40)):
                data = b""
                while data[:1] not in (b">", b"@"):
                    data = fasta_file(rng) if rng.random() < 0.6 else fastq_file(rng)
                if rng.random() < 0.2:                       # larger files -> several batches at --batch-mb 1
                    if not data.endswith(b"\n"):
                        data += b"\n"                       # (a FASTQ glued mid-line would be malformed anyway)
                    data = data * rng.randint(2, 30)
                plain = os.path.join(td, "p%d_%d" % (it, i))
                open(plain, "wb").write(data)
                recs.append(H.read_fastx(plain))
                comp = rng.choice(["", "", "gz", "bz2", "xz", "zst"])
                path = plain + ("." + comp if comp else "")
                if comp == "gz":
                    half = len(data) // 2                    # two concatenated gzip members
                    open(path, "wb").write(gzip.compress(data[:half], 1) + gzip.compress(data[half:], 1))
                elif comp == "bz2":
                    open(path, "wb").write(bz2.compress(data, 1))
                elif comp == "xz":
                    open(path, "wb").write(lzma.compress(data, format=lzma.FORMAT_XZ, preset=0))
                elif comp == "zst":
                    H.zstd_write(path, data, level=1, workers=0)
                paths.append(path)
            lst = os.path.join(td, "l.txt")
            open(lst, "w").write("\n".join(paths) + "\n")
            algo = rng.choice(["hmh", "hll", "ull"])
            if os.environ.get("FUZZ_ALGO"):
                algo = os.environ["FUZZ_ALGO"]
            k = rng.choice([rng.randint(1, 32), 16, 21])
            p = rng.randint(4, 14)
            if os.environ.get("FUZZ_P") and algo != "hmh":           # pin the precision (register tables beyond LDS: tools/bins_fuzz.sh)
                p = min(rng.choice([int(x) for x in os.environ["FUZZ_P"].split(",")]), 16 if algo == "hll" else 26)
            extra = rng.choice([[], ["--batch-mb", "1"], ["--batch-mb", "1", "--devices", "0,0"], ["--devices", "0,0,0"], ["-t", "2"]])
            out = os.path.join(td, "o")
            r = subprocess.run([H.CLI, "sketch", "-f", lst, "-o", out, "-a", algo, "-k", str(k), "-p", str(p)] + extra, capture_output=True, text=True)
            if r.returncode != 0:
                print("FAILED it=%d: %s" % (it, r.stderr[-400:]))
                sys.exit(1)
            blob = H.zstd_read(out + "_sketches.bin")
            seq, off, goff = lash_amd.records_to_arrays(recs)
            want = O.sketch_genomes(ALGO[algo], k, p if algo != "hmh" else 0, 42, seq, off, goff, threads=8)
            if blob != want.tobytes():
                ib = want.shape[1]
                bad = [i for i in range(len(paths)) if blob[i * ib:(i + 1) * ib] != want[i].tobytes()]
                print("MISMATCH [" + knobs + "] it=%d %s k=%d p=%d extra=%s files=%s" % (it, algo, k, p, extra, [paths[i] for i in bad][:5]))
                os.system("cp %s /tmp/ 2>/dev/null" % " ".join(paths[i] for i in bad[:3]))
                sys.exit(1)
            for f in os.listdir(td):
                os.remove(os.path.join(td, f))
    print("cli fuzz ok: %d iterations from seed %d" % (iters, seed0))


if __name__ == "__main__":
    main()
