#!/usr/bin/env python3
"""tests/fuzz_gpu_stream.py [iterations] [seed] — the CLI's large-file streaming (record-aligned chunks accumulated on
the device) against the same files sketched whole: `lash sketch --stream-mb 1` vs the default.  GPU box, manual."""
import os
import random
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import host_lib as H
import oracle_lib as O
import fuzz_knobs


def big_fasta(rng):
    g = O.synth_genome(rng.randint(0, 10**6), 6_000_000).tobytes()
    out, pos = [], 0
    target = rng.randint(2_500_000, 5_500_000)
    while pos < target:
        n = rng.choice([rng.randint(1, 2000), rng.randint(100_000, 900_000), rng.randint(1_200_000, 2_600_000)])
        n = min(n, target - pos)
        s = bytearray(g[pos:pos + n])
        if rng.random() < 0.5:
            i = rng.randrange(n)
            run = rng.choice([500, 5000, 50_000, 400_000])          # N runs far longer than any fixed overlap
            s[i:i + run] = b"N" * len(s[i:i + run])
        w = rng.choice([60, 80, 80, 10**9, 7])
        out.append(b">r%d %s\n" % (len(out), b"x" * rng.choice([0, 10, 5000])))
        out.append(b"\n".join(bytes(s[i:i + w]) for i in range(0, n, w)) + b"\n")
        pos += n
    return b"".join(out)


def big_fastq(rng):
    g = O.synth_genome(rng.randint(0, 10**6), 2_000_000).tobytes()
    out = []
    size = 0
    target = rng.randint(2_500_000, 4_500_000)
    while size < target:
        L = rng.choice([100, 150, 151, rng.randint(1, 1000)])
        s = rng.randrange(0, len(g) - L)
        rec = b"@r%d\n" % len(out) + g[s:s + L] + b"\n+\n" + bytes(rng.choice(b"@+IIFF>#") for _ in range(L)) + b"\n"
        out.append(rec)
        size += len(rec)
    return b"".join(out)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        for it in range(iters):
            rng = random.Random(seed0 * 104729 + it)
            knobs = fuzz_knobs.set_sole(random.Random(seed0 * 1000003 + it))    # which genomes go to the persistent small-genome kernel (FUZZ_SOLE)
            paths = []
            for i in range(rng.randint(1, 3)):
                data = big_fasta(rng) if rng.random() < 0.6 else big_fastq(rng)
                path = os.path.join(td, "f%d_%d" % (it, i))
                open(path, "wb").write(data)
                paths.append(path)
            small = os.path.join(td, "small_%d.fa" % it)
            open(small, "wb").write(b">s\nACGTACGTTTGACCA\n")
            paths.insert(rng.randint(0, len(paths)), small)
            lst = os.path.join(td, "l.txt")
            open(lst, "w").write("\n".join(paths) + "\n")
            algo, k, p = rng.choice([("hmh", 16, 10), ("hll", 21, 12), ("ull", 31, 10), ("hmh", 7, 10)])
            blobs = []
            for tag, extra in (("whole", []), ("stream", ["--stream-mb", "1"])):
                out = os.path.join(td, tag)
                r = subprocess.run([H.CLI, "sketch", "-f", lst, "-o", out, "-a", algo, "-k", str(k), "-p", str(p), "-t", "4"] + extra,
                                   capture_output=True, text=True)
                if r.returncode != 0:
                    print("FAILED it=%d %s: %s" % (it, tag, r.stderr[-500:]))
                    sys.exit(1)
                blobs.append(H.zstd_read(out + "_sketches.bin"))
            if blobs[0] != blobs[1]:
                for pth in paths:
                    os.system("cp %s /tmp/" % pth)
                print("MISMATCH [" + knobs + "] it=%d %s k=%d (inputs copied to /tmp)" % (it, algo, k))
                sys.exit(1)
            for pth in paths:
                os.remove(pth)
    print("stream fuzz ok: %d iterations from seed %d" % (iters, seed0))


if __name__ == "__main__":
    main()
