#!/usr/bin/env python3
"""Decompress what the real `lash` wrote and file it under tests/golden/ref_images/ with a sha256 manifest."""
import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def zstd_decompress(path):
    try:
        import zstandard
        with open(path, "rb") as f:
            return zstandard.ZstdDecompressor().stream_reader(f).read()
    except ImportError:
        return subprocess.check_output(["zstd", "-dc", path])


def sorted_rows(path):
    """`lash dist` TSV -> header + rows sorted by (reference, query): the reference's row order is nondeterministic"""
    lines = open(path).read().splitlines()
    if not lines:
        return ""
    return "\n".join([lines[0]] + sorted(lines[1:])) + "\n"


def main(work, out):
    sys.path.insert(0, HERE)
    from make_inputs import ERROR_INPUTS, INPUTS, ORDER_NAMES
    cases = []
    for line in open(os.path.join(HERE, "cases.tsv")):
        if line.startswith("#") or not line.strip():
            continue
        name, algo, k, p, seed = line.rstrip("\n").split("\t")
        d = os.path.join(work, name)
        raw = zstd_decompress(os.path.join(d, name + "_sketches.bin"))
        assert len(raw) % len(INPUTS) == 0, "%s: %d bytes is not a multiple of %d sketches" % (name, len(raw), len(INPUTS))
        with open(os.path.join(out, name + ".bin"), "wb") as f:
            f.write(raw)
        entry = {"name": name, "algo": algo, "k": int(k), "p": int(p), "seed": int(seed), "images": name + ".bin",
                 "image_bytes": len(raw) // len(INPUTS), "sha256": hashlib.sha256(raw).hexdigest()}
        for suffix, key in ((".dist.raw", "dist"), (".dist_ml.raw", "dist_ml")):
            src = os.path.join(d, name + suffix)
            if os.path.exists(src):
                txt = sorted_rows(src)
                dst = name + suffix.replace(".raw", ".tsv")
                with open(os.path.join(out, dst), "w") as f:
                    f.write(txt)
                entry[key] = {"tsv": dst, "sha256": hashlib.sha256(txt.encode()).hexdigest()}
        for js in ("_files.json", "_parameters.json"):
            shutil.copy(os.path.join(d, name + js), os.path.join(out, name + js))
        cases.append(entry)
    version = open(os.path.join(work, "lash_version.txt")).read().strip() if os.path.exists(os.path.join(work, "lash_version.txt")) else ""
    manifest = {"generated_by": "tools/ref_probe/run.sh", "lash_version": version,
                "crates": {"kmerutils": "0.0.14", "hyperminhash": "0.1.4", "streaming_algorithms": "0.3.3",
                           "ultraloglog": "0.1.6", "xxhash-rust": "0.8.15", "needletail": "0.6.3"},
                "inputs": INPUTS, "inputs_dir": "tests/golden (appendix_b.fasta: '>appendix_b' + ACGTTGCATGCATCGATCGGATTACA)",
                "cases": cases}
    # the map-order probe, verbatim (its order IS the datum)
    od = os.path.join(work, "order")
    if os.path.exists(os.path.join(od, "order.dm.raw")):
        manifest["order"] = {"names": ORDER_NAMES}
        for src, key in (("order.dm.raw", "dm"), ("order.rows.raw", "rows")):
            if os.path.exists(os.path.join(od, src)):
                shutil.copy(os.path.join(od, src), os.path.join(out, src.replace(".raw", ".txt")))
                manifest["order"][key] = src.replace(".raw", ".txt")
    # the error probe: the images of the two FASTQ files with a malformed record in the middle (hmh k=16 seed 42), or the fact
    # that the real lash did not get that far
    ed = os.path.join(work, "errors")
    if os.path.isdir(ed):
        binf = os.path.join(ed, "errors_sketches.bin")
        if os.path.exists(binf):
            raw = zstd_decompress(binf)
            with open(os.path.join(out, "errors.bin"), "wb") as f:
                f.write(raw)
            manifest["errors"] = {"inputs": ERROR_INPUTS, "images": "errors.bin", "sha256": hashlib.sha256(raw).hexdigest(),
                                  "algo": "hmh", "k": 16, "seed": 42}
        else:
            log = open(os.path.join(ed, "sketch.log")).read()[-2000:] if os.path.exists(os.path.join(ed, "sketch.log")) else ""
            manifest["errors"] = {"inputs": ERROR_INPUTS, "failed": True, "log_tail": log}
    with open(os.path.join(out, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote %d cases to %s" % (len(cases), out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
