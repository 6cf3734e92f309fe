"""tools/ref_probe/extract_hll_bias.py finds the HLL++ tables by shape: feed it Rust-looking sources with SYNTHETIC
numbers in the styles a crate may use, and check that the file it writes loads and round-trips."""
import os
import subprocess
import sys

import numpy as np
import pytest

import lash_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "ref_probe", "extract_hll_bias.py")


def _tables(seed=1):
    rng = np.random.default_rng(seed)
    raw, bias = [], []
    for p in range(4, 19):
        m = float(1 << p)
        n = 80 if p == 4 else 160 if p == 5 else 200
        r = np.round(np.sort(rng.uniform(0.7 * m, 5 * m, n)), 4)
        r[0] = round(0.7 * m, 4)
        b = np.round(0.65 * m * np.exp(-(r - 0.7 * m) / m) + rng.normal(0, 0.003 * m, n), 4)
        raw.append(r.tolist())
        bias.append(b.tolist())
    return raw, bias


def _rust(raw, bias, style):
    fmt = (lambda x: repr(x)) if style != "suffix" else (lambda x: repr(x) + "_f64")
    out = ["// synthetic stand-in for a crate's constants file", "pub const THRESHOLD_DATA: [f64; 15] = [10.0, 20.0, 40.0, 80.0, 220.0, 400.0, 900.0, 1800.0, 3100.0, 6500.0, 11500.0, 20000.0, 50000.0, 120000.0, 350000.0];"]
    if style == "separate":
        for name, t in (("RAW", raw), ("BIAS", bias)):
            for i, row in enumerate(t):
                out.append("const %s_%d: [f64; %d] = [ %s ]; /* p = %d */" % (name, 4 + i, len(row), ", ".join(fmt(x) for x in row), 4 + i))
    else:
        for name, t in (("RAW_ESTIMATE_DATA", raw), ("BIAS_DATA", bias)):
            out.append("pub const %s: &[&[f64]] = &[" % name)
            for i, row in enumerate(t):
                out.append("    // precision %d" % (4 + i))
                out.append("    &[" + ",\n      ".join(", ".join(fmt(x) for x in row[j:j + 8]) for j in range(0, len(row), 8)) + ",],")
            out.append("];")
    out.append("fn unrelated() { let v = [1, 2, 3]; let w = vec![0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5]; }")
    return "\n".join(out) + "\n"


@pytest.mark.parametrize("style", ["nested", "suffix", "separate"])
def test_extractor_finds_tables_by_shape(tmp_path, style):
    raw, bias = _tables()
    crate = tmp_path / "streaming_algorithms-0.3.3" / "src" / "distinct"
    crate.mkdir(parents=True)
    (crate / "consts.rs").write_text(_rust(raw, bias, style))
    (crate / "mod.rs").write_text("pub fn len() -> f64 { let x = [0u8; 16]; 0.0 }\n")
    out = tmp_path / "hllpp_bias.txt"
    r = subprocess.run([sys.executable, TOOL, str(tmp_path / "streaming_algorithms-0.3.3"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {}
    cur = None
    for ln in out.read_text().splitlines():
        if ln.startswith("#"):
            continue
        if ln.startswith("p "):
            cur = int(ln.split()[1]); got[cur] = ([], [])
        else:
            a, b = ln.split(); got[cur][0].append(float(a)); got[cur][1].append(float(b))
    assert sorted(got) == list(range(4, 19))
    for i in range(15):
        assert got[4 + i] == (raw[i], bias[i])
    tb = lash_amd.HllBias(out)                      # and the library reads what the tool wrote
    assert all(tb.has(p) for p in range(4, 19))


def test_extractor_says_so_when_there_is_nothing(tmp_path):
    (tmp_path / "lib.rs").write_text("pub fn f() -> [f64; 3] { [1.0, 2.0, 3.0] }\n")
    r = subprocess.run([sys.executable, TOOL, str(tmp_path), str(tmp_path / "o.txt")], capture_output=True, text=True)
    assert r.returncode == 3 and not (tmp_path / "o.txt").exists()
