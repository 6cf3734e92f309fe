#!/usr/bin/env python3
"""tests/fuzz_gpu_dist.py [iterations] [seed] — `lash dist` (HyperMinHash, HyperLogLog, UltraLogLog FGRA / ML) on random small genome sets and
random output options against the pure-Python restatement in tests/pyref.py.  GPU box, manual."""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import host_lib as H
import oracle_lib as O
import pyref as R


def mutated(seq, rate, rng):
    out = seq.copy()
    idx = rng.random(len(seq)) < rate
    out[idx] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(idx.sum()))
    return out


WORST = {}


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for it in range(iters):
        rng = random.Random(seed0 * 49979687 + it)
        nrng = np.random.default_rng(seed0 * 1000 + it)
        algo = rng.choice(["hmh", "hll", "ull"])
        est = rng.choice(["fgra", "ml"])
        k = rng.choice([16, 21, 12, 31])
        p = rng.randint(8, 14)
        L = rng.choice([300_000, 600_000])
        base = O.synth_genome(rng.randint(0, 10**6), L)
        genomes = [base] + [mutated(base, rng.choice([0.0005, 0.005, 0.05, 0.3]), nrng) for _ in range(rng.randint(1, 4))] + [O.synth_genome(rng.randint(0, 10**6), L)]
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            paths = []
            for i, g in enumerate(genomes):
                path = os.path.join(td, "g%d.fa" % i)
                open(path, "wb").write(b">g\n" + g.tobytes() + b"\n")
                paths.append(path)
            nq = rng.randint(1, len(paths))
            open(os.path.join(td, "r.txt"), "w").write("\n".join(paths) + "\n")
            open(os.path.join(td, "q.txt"), "w").write("\n".join(paths[:nq]) + "\n")
            for pre, lst in (("refs", "r.txt"), ("qry", "q.txt")):
                r = subprocess.run([H.CLI, "sketch", "-f", os.path.join(td, lst), "-o", pre, "-k", str(k), "-a", algo, "-p", str(p)], cwd=td, capture_output=True, text=True)
                assert r.returncode == 0, r.stderr
            algo_id = O.HMH if algo == "hmh" else O.HLL if algo == "hll" else O.ULL
            imgs = [O.sketch_genomes(algo_id, k, p if algo != "hmh" else 0, 42, g, np.array([0, len(g)], np.uint64), np.array([0, 1], np.uint64))[0].tobytes() for g in genomes]
            model = rng.choice([0, 1])
            fp32 = rng.random() < 0.3
            same = rng.random() < 0.5
            matrix = same and rng.random() < 0.5
            q = "refs" if same else "qry"
            cmd = [H.CLI, "dist", "-q", q, "-r", "refs", "-o", "d.txt", "-m", str(model), "-t", str(rng.choice([1, 3, 8]))]
            if algo == "ull":
                cmd += ["-e", est]
                estimate = R.ull_fgra if est == "fgra" else R.ull_ml
                card = [estimate(list(im[8:]), p) for im in imgs]
            cmd += (["--fp32"] if fp32 else []) + (["--dm"] if matrix else []) + rng.choice([[], ["--block-rows", "2"]])
            file_order = rng.random() < 0.4
            cmd += ["--file-order"] if file_order else []
            # rows / columns: list-file order, or the reference's seeded hash-map key order (name_order.hpp)
            ro = list(range(len(paths))) if file_order else R.hashbrown_name_order(paths)
            qo = ro if same else list(range(nq)) if file_order else R.hashbrown_name_order(paths[:nq])
            r = subprocess.run(cmd, cwd=td, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            text = open(os.path.join(td, "d.txt")).read()

            def expected(ri, qi):
                if algo == "ull":
                    u = estimate(list(R.ull_merge(imgs[ri][8:], imgs[qi][8:])), p)
                    sim = (card[ri] + card[qi] - u) / u
                else:
                    sim = R.hmh_similarity(imgs[qi], imgs[ri]) if algo == "hmh" else R.hll_similarity(p, imgs[ri], imgs[qi])
                return R.mash_distance(sim, k, model, paths[ri] == paths[qi])

            tol = 3e-6 if fp32 else 1.1e-6
            got = {}
            if matrix:
                lines = text.split("\n")
                assert lines[0] == "".join("\t" + paths[j] for j in qo), "matrix header"
                for a, ln in enumerate(lines[1:]):
                    cells = ln.split("\t")
                    assert cells[0] == paths[ro[a]] and len(cells) == a + 2
                    for b, d in enumerate(cells[1:]):
                        got[(ro[a], qo[b])] = float(d)
                want_pairs = [(ro[a], ro[b]) for a in range(len(paths)) for b in range(a + 1)]
            else:
                lines = text.strip().split("\n")
                assert lines[0] == "Reference\tQuery\tDistance"
                for ln in lines[1:]:
                    a, b, d = ln.split("\t")
                    got[(paths.index(a), paths.index(b))] = float(d)
                want_pairs = [(ro[a], ro[b]) for a in range(len(paths)) for b in range(a + 1)] if same else [(i, j) for i in ro for j in qo]
                if [(paths.index(ln.split("\t")[0]), paths.index(ln.split("\t")[1])) for ln in lines[1:]] != want_pairs:
                    print("MISMATCH it=%d: row order differs" % it)
                    sys.exit(1)
            if sorted(got) != sorted(want_pairs):
                print("MISMATCH it=%d: pair set differs: %s vs %s" % (it, sorted(got), sorted(want_pairs)))
                sys.exit(1)
            for (i, j) in want_pairs:
                e = expected(i, j)
                key = algo + ("-" + est if algo == "ull" else "")
                WORST[key] = max(WORST.get(key, 0.0), abs(got[(i, j)] - e))
                if abs(got[(i, j)] - e) > tol:
                    print("MISMATCH it=%d %s k=%d p=%d model=%d fp32=%s pair=(%d,%d): %r vs %r" % (it, algo, k, p, model, fp32, i, j, got[(i, j)], e))
                    sys.exit(1)
    print("largest |CLI - restatement| per estimator:", {k: "%.2e" % v for k, v in sorted(WORST.items())})
    print("dist fuzz ok: %d iterations from seed %d" % (iters, seed0))


if __name__ == "__main__":
    main()
