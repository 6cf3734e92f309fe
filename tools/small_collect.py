#!/usr/bin/env python3
"""tools/small_collect.py gpurun_out/<tag>/small profiles/rNN/small — file the rocprofv3 summaries of the small-genome shapes
(tools/profile_py.sh on tools/one_shape.py / tools/viral_rate.py) and write the roofline line of each: algorithmic bytes per launch =
sum over genomes of (record bytes L + image bytes S) — what SURVEY §8(d) counts: every input byte read once, every image written once —
over the persistent kernel's average launch time from rocprofv3's kernel stats, against 8 TB/s; HBM bytes from the PMC passes
(2 x FETCH_SIZE + WRITE_SIZE, KiB; MI355X_MICROARCH.md's gfx950 correction); vector instructions per k-mer from SQ_INSTS_VALU."""
import csv
import os
import re
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_collect import parse_pmc

IMG = {"hmh": 32768, "hll10": 33 + 1024}
SHAPES = {   # name -> [(kernel prefix, genomes, bytes of sequence, k-mers, image bytes per genome)]
    "hmh_100000x10k": [("lash::sole_sketch_kernel<0,", 100000, 100000 * 10000, 100000 * (10000 - 15), IMG["hmh"])],
    "hll_p10_100000x10k": [("lash::sole_sketch_kernel<1,", 100000, 100000 * 10000, 100000 * (10000 - 20), IMG["hll10"])],
    "hmh_1000000x1k": [("lash::sole_sketch_kernel<0,", 1000000, 1000000 * 1000, 1000000 * (1000 - 15), IMG["hmh"])],
    "hll_p10_1000000x1k": [("lash::sole_sketch_kernel<1,", 1000000, 1000000 * 1000, 1000000 * (1000 - 20), IMG["hll10"])],
    "hmh_20000x50k": [("lash::sole_sketch_kernel<0,", 20000, 20000 * 50000, 20000 * (50000 - 15), IMG["hmh"])],
}


def viral(run_log):
    """tools/viral_rate.py prints genomes, GB and the wall rate; k-mers = rate x wall"""
    t = open(run_log).read()
    m = re.search(r"(\d+) genomes, ([0-9.]+) GB, (\d+) records", t)
    g, gb = int(m.group(1)), float(m.group(2)) * 1e9
    out = []
    for an, pre, img in (("hmh", "lash::sole_sketch_kernel<0,", IMG["hmh"]), ("hll", "lash::sole_sketch_kernel<1,", IMG["hll10"])):
        m = re.search(an + r" k=\d+ p=\d+: wall ([0-9.]+) ms per call.*-> ([0-9.e+]+) k-mers/s", t)
        out.append((pre, g, gb, float(m.group(2)) * float(m.group(1)) * 1e-3, img))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith(".txt"):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    rows = []
    for name in sorted(os.listdir(src)):
        d = os.path.join(src, name)
        if not os.path.isdir(d):
            continue
        os.makedirs(os.path.join(dst, name), exist_ok=True)
        for f in ("kernel_stats.csv", "pmc_summary.txt", "run.log"):
            if os.path.exists(os.path.join(d, f)):
                shutil.copy(os.path.join(d, f), os.path.join(dst, name, f))
        shapes = viral(os.path.join(d, "run.log")) if name.startswith("viral") else SHAPES.get(name, [])
        pmc = parse_pmc(os.path.join(d, "pmc_summary.txt"))
        stats = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
        for pre, g, seq_bytes, kmers, img in shapes:
            r = next((x for x in stats if x["Name"].replace("void ", "").startswith(pre)), None)
            c = next((v for kname, v in pmc.items() if kname.startswith(pre)), {})
            if r is None:
                continue
            ms = float(r["AverageNs"]) / 1e6
            alg = seq_bytes + g * img
            fetch, write, insts = c.get("FETCH_SIZE"), c.get("WRITE_SIZE"), c.get("SQ_INSTS_VALU")
            hbm = 2 * fetch * 1024 + write * 1024 if fetch is not None and write is not None else None
            act, busy = c.get("SQ_ACTIVE_INST_VALU"), c.get("SQ_BUSY_CYCLES")
            rows.append((name, r["Name"].split("(")[0].replace("void ", ""), int(r["Calls"]), ms, g, alg, alg / (ms * 1e-3) / 8e12, hbm,
                         insts / (kmers / 64.0) if insts else None, kmers / (ms * 1e-3)))
    with open(os.path.join(dst, "SUMMARY.md"), "w") as f:
        f.write("# %s — the persistent small-genome kernel under rocprofv3 (tools/r05_profiles.sh; one MI355X)\n\n" % dst +
                "Algorithmic bytes per launch = Σ over genomes of (record bytes L + image bytes S): SURVEY §8(d)'s accounting — every input byte read once, every image written once.\n"
                "`roofline`: bound hbm, achieved = algorithmic bytes ÷ rocprofv3's average launch time, peak 8 TB/s.  The binding roofline is vector issue (DESIGN §4.6): the last\n"
                "two columns.  `*.txt` beside this file: `tools/small_genomes_rate.py` / `tools/viral_rate.py` on the wall clock, with the persistent kernel and (`*_sliced_kernels.txt`,\n"
                "`LASH_SOLE_MAX=0`) with round 4's route on the same box.\n\n"
                "| shape | kernel | launches | avg ms | genomes | algorithmic bytes | achieved TB/s | **frac of 8 TB/s** | HBM bytes (PMC) | traffic / algorithmic | VALU wave-instr per 64 k-mers | k-mers/s |\n"
                "|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for n, kern, calls, ms, g, alg, frac, hbm, ipk, rate in rows:
            f.write("| %s | `%s` | %d | %.3f | %d | %.4g | %.3f | **%.4f** | %s | %s | %s | %.3e |\n" % (
                n, kern, calls, ms, g, alg, alg / (ms * 1e-3) / 1e12, frac, "%.4g" % hbm if hbm else "-", "%.3f" % (hbm / alg) if hbm else "-",
                "%.1f" % ipk if ipk else "-", rate))
    print(open(os.path.join(dst, "SUMMARY.md")).read())


if __name__ == "__main__":
    main()
