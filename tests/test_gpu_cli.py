"""GPU test of the drop-in surface: `lash sketch` (C++ CLI over the C ABI) writes {o}_sketches.bin / {o}_files.json /
{o}_parameters.json exactly as the reference does (utils.rs:567-580, main.rs:254-276); the decompressed .bin must be
the concatenation of the oracle's images in list order (BASELINE.json configs[0] plumbing on the bundled fixtures)."""
import json
import os
import subprocess

import numpy as np
import pytest

import host_lib as H
import oracle_lib as O
from fastx import read_fastx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGO = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}


@pytest.mark.parametrize("algo,k,p,extra", [("hmh", 16, 10, []), ("hll", 21, 14, []), ("ull", 16, 12, ["--batch-mb", "1"]),
                                            ("hmh", 11, 10, ["-t", "3", "--batch-mb", "1"])])
def test_lash_sketch_cli_outputs(tmp_path, algo, k, p, extra):
    names = ["fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq", "fixture_A.fasta"]
    paths = [os.path.join(GOLD, n) for n in names]
    lst = tmp_path / "genomes.txt"
    lst.write_text("\n".join(paths) + "\n\n")
    out = str(tmp_path / "sk")
    cmd = [H.CLI, "sketch", "-f", str(lst), "-o", out, "-a", algo, "-k", str(k), "-p", str(p), "-s", "42"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.load(open(out + "_files.json")) == paths
    assert open(out + "_files.json").read() == H.json_array(paths)
    params = json.load(open(out + "_parameters.json"))
    want_params = {"algorithm": algo, "k": str(k), "molecule": "nucleotide", "seed": "42"}
    if algo != "hmh":
        want_params["precision"] = str(p)
    assert params == want_params
    blob = H.zstd_read(out + "_sketches.bin")
    ib = O.image_bytes(ALGO[algo], p)
    assert len(blob) == ib * len(paths)
    for i, path in enumerate(paths):
        recs = read_fastx(path)
        seq = np.frombuffer(b"".join(recs), np.uint8)
        off = np.cumsum([0] + [len(x) for x in recs]).astype(np.uint64)
        want = O.sketch_genomes(ALGO[algo], k, p, 42, seq, off, np.array([0, len(recs)], np.uint64))[0].tobytes()
        assert blob[i * ib:(i + 1) * ib] == want, (algo, names[i])
