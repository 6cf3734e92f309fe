#!/usr/bin/env python3
"""tools/profile_collect.py gpurun_out/<tag> [profiles/rNN] — file the rocprofv3 summaries of tools/rNN_profiles.sh under
profiles/rNN/<name>/ and recompute, per workload, what bench.py's roofline fields quote:
  profiles/traffic.json   HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the x2 is
                          the gfx950 correction of MI355X_MICROARCH.md's HBM section, re-checked by the calibration run)
  profiles/valu.json      SQ_INSTS_VALU per k-mer of the dominant kernel + the measured issue ceiling (tools/ubench_hash)
and prints the table DESIGN.md §5 carries (kernel, avg launch ms under rocprofv3, algorithmic bytes, frac of 8 TB/s, traffic)."""
import csv
import json
import os
import re
import shutil
import sys

HBM_PEAK = 8000.0


def parse_pmc(path):
    """pmc_summary.txt -> {kernel: {counter: avg, 'dispatches': n, 'avg_ns': ns}} (last block per kernel+pass wins per counter)"""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"(\S+)\s+(.+?)\s+dispatches=(\d+) avg_ns=(\d+)", line)
        if m:
            cur = out.setdefault(m.group(2).strip(), {})
            cur.setdefault("passes", {})[m.group(1)] = (int(m.group(3)), int(m.group(4)))
            continue
        m = re.match(r"\s+(\S+)\s+avg/dispatch = (\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02"
    os.makedirs(dst, exist_ok=True)
    rnd = int(re.search(r"r(\d+)", os.path.basename(dst.rstrip("/"))).group(1))
    traffic_path, valu_path = "profiles/traffic.json", "profiles/valu.json"
    traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    valu = json.load(open(valu_path)) if os.path.exists(valu_path) else {}
    floors = {}
    ub = os.path.join(src, "ubench_hash.txt")
    if os.path.exists(ub):
        shutil.copy(ub, os.path.join(dst, "ubench_hash.txt"))
        txt = open(ub).read()
        for key, pat in ((("hmh", 16), r"\+ ds_max_u32"), (("hll", 21), "hll p14 k21 stream"), (("ull", 16), "ull p12 k16 stream"),
                         (("hmh", 16, "defer"), "defer: hmh k16 stream")):
            m = re.search(pat + r".*\(([0-9.e+]+) k-mers/s chip-wide\)", txt)
            if m:
                floors[key] = float(m.group(1))
    cal = os.path.join(src, "calibration", "calibration.txt")
    if os.path.exists(cal):
        shutil.copy(cal, os.path.join(dst, "hbm_calibration.txt"))
    rows = []
    for name in sorted(os.listdir(src)):
        d = os.path.join(src, name)
        bj = os.path.join(d, "bench.json")
        if not os.path.isdir(d) or not os.path.exists(bj):
            continue
        os.makedirs(os.path.join(dst, name), exist_ok=True)
        for f in ("bench.json", "kernel_stats.csv", "pmc_summary.txt"):
            if os.path.exists(os.path.join(d, f)):
                shutil.copy(os.path.join(d, f), os.path.join(dst, name, f))
        try:
            j = json.loads(open(bj).read().strip().splitlines()[-1])
        except Exception as e:
            print(name, "no bench line:", e)
            continue
        cfg, roof = j["config"], j["roofline"]
        direct = "DIRECT" in roof["kernel"]
        pmc = parse_pmc(os.path.join(d, "pmc_summary.txt")) if os.path.exists(os.path.join(d, "pmc_summary.txt")) else {}
        # the dominant kernel = the sketch_kernel instantiation with the largest total time in the stats
        dom, dom_ms, dom_calls = None, None, None
        ks = os.path.join(d, "kernel_stats.csv")
        if os.path.exists(ks):
            best = 0
            for r in csv.DictReader(open(ks)):
                if "sketch_kernel" in r["Name"] and float(r["TotalDurationNs"]) > best:
                    best = float(r["TotalDurationNs"])
                    dom, dom_ms, dom_calls = r["Name"], float(r["AverageNs"]) / 1e6, int(r["Calls"])
        short = dom.split("(")[0].replace("void ", "") if dom else None
        c = pmc.get(short, {}) if short else {}
        if short and not c:                             # (pmc_summary.py prints the first 50 characters of a kernel's name)
            c = next((v for kname, v in pmc.items() if len(kname) >= 40 and short.startswith(kname)), {})
        fetch, write, insts = c.get("FETCH_SIZE"), c.get("WRITE_SIZE"), c.get("SQ_INSTS_VALU")
        hbm = int(2 * fetch * 1024 + write * 1024) if fetch is not None and write is not None else None
        kmers = cfg["genomes_per_gpu"] * (cfg["genome_length"] - cfg["k"] + 1) if cfg.get("records_per_gpu", 0) == cfg["genomes_per_gpu"] \
            else cfg["records_per_gpu"] * (150 - cfg["k"] + 1)
        if cfg.get("dirty") == "nrun":
            kmers = cfg["genomes_per_gpu"] * (cfg["genome_length"] - 100 - cfg["k"] + 1)
        elif cfg.get("dirty") == "lower":
            kmers = cfg["genomes_per_gpu"] * (cfg["genome_length"] // 2 - cfg["k"] + 1)
        key = "%s%s_k%d_p%d_g%d_l%d" % ("direct_" if direct else "", cfg["algo"], cfg["k"], cfg["p"], cfg["genomes_per_gpu"], cfg["genome_length"])
        default_layout = cfg.get("layout", "default") == "default"      # (bench.py --layout runs are listed, but the committed figures bench.py quotes are the default rule's)
        if hbm is not None and cfg.get("dirty", "none") == "none" and default_layout:
            traffic[key] = {"hbm_bytes_per_launch": hbm, "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
                            "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE x1; calibrated in the same session (%s/hbm_calibration.txt)" % dst,
                            "source": "%s/%s/pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes)" % (dst, name),
                            "kernel": short, "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"], "round": rnd}
        defer = "DEFER" in roof["kernel"]               # HyperMinHash with deferred signatures: a stream, a count and a ceiling of its own
        vkey = "%s%s_k%d%s" % ("direct_" if direct else "", cfg["algo"], cfg["k"], "_defer" if defer else "")
        if insts is not None and cfg.get("dirty", "none") == "none" and default_layout:
            # one entry per kernel: the deferring kernel's from the default workload (whole genomes: what bench.py runs by default), the
            # others' from the 1 000-genome / read-set runs unless only a large run exists
            if (cfg["genomes_per_gpu"] == 12500 or vkey not in valu) if defer else \
               (cfg["genomes_per_gpu"] not in (10000, 12500) or vkey not in valu or valu[vkey].get("round", 0) < rnd):
                valu[vkey] = {"valu_insts_per_kmer": insts / (kmers / 64.0),
                              "note": "SQ_INSTS_VALU counts wave-instructions: per k-mer = SQ_INSTS_VALU / (k-mers / 64)",
                              "source": "%s/%s/pmc_summary.txt" % (dst, name), "round": rnd}
                fkey = (cfg["algo"], cfg["k"], "defer") if defer else (cfg["algo"], cfg["k"])
                if fkey in floors:
                    valu[vkey]["issue_floor_kmers_per_s"] = floors[fkey]
                    valu[vkey]["floor_source"] = "%s/ubench_hash.txt (tools/ubench_hash, same session)" % dst
        alg = roof["algorithmic_bytes_per_launch"]
        rows.append((name, short, dom_calls, dom_ms, roof["avg_launch_ms"], alg, alg / (dom_ms * 1e-3) / 1e9 / HBM_PEAK if dom_ms else None,
                     hbm, insts / (kmers / 64.0) if insts else None, j["value"]))
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    json.dump(valu, open(valu_path, "w"), indent=1)
    with open(os.path.join(dst, "SUMMARY.md"), "w") as f:
        f.write("# %s — rocprofv3 summaries per workload (tools/r%02d_profiles.sh, one MI355X)\n\n" % (dst, rnd) +
                "| workload | dominant kernel | launches | avg ms (rocprofv3) | avg ms (HIP events, unprofiled run) | algorithmic bytes | frac of 8 TB/s | HBM bytes (PMC) | VALU wave-instr per 64 k-mers | k-mers/s |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | `%s` | %s | %s | %.3f | %.4g | %s | %s | %s | %.3e |\n" % (
                r[0], r[1], r[2], "%.3f" % r[3] if r[3] else "-", r[4], r[5], "%.4f" % r[6] if r[6] else "-",
                "%.4g" % r[7] if r[7] else "-", "%.1f" % r[8] if r[8] else "-", r[9]))
    print(open(os.path.join(dst, "SUMMARY.md")).read())


if __name__ == "__main__":
    main()
