"""CPU tests of the drop-in boundary: liblash_gfx950.so loads without a GPU, exports every function
include/lash_gfx950.h declares, and its host-only entry points (sizes, parameter checks, error text) behave like the
reference's panics/expects.  No compute calls: there is no GPU here and no CPU fallback in the product."""
import ctypes as C
import os
import re

import pytest

import lash_amd
from lash_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "lash_gfx950.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lash_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_functions()
    assert len(names) >= 20
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "liblash_gfx950.so does not export %s" % n
        assert n in _lib.PROTOTYPES, "lash_amd/_lib.py has no prototype for %s" % n
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.lash_abi_version() == 5


def test_image_sizes_and_param_checks():
    assert lash_amd.image_bytes("hmh") == 32768
    assert lash_amd.image_bytes("hll", 14) == 33 + 16384
    assert lash_amd.image_bytes("ull", 12) == 8 + 4096
    assert lash_amd.image_bytes("hll", 3) == 0 and lash_amd.image_bytes("ull", 27) == 0
    ok = [("hmh", 1, 0), ("hmh", 32, 99), ("hll", 16, 4), ("hll", 16, 16), ("ull", 16, 3), ("ull", 16, 26)]
    bad = [("hmh", 0, 0), ("hmh", 33, 0), ("hll", 16, 3), ("hll", 16, 17), ("ull", 16, 2), ("ull", 16, 27)]
    for a, k, p in ok:
        assert lash_amd.params_check(a, k, p) == lash_amd.OK, (a, k, p)
    for a, k, p in bad:
        assert lash_amd.params_check(a, k, p) == lash_amd.EINVAL, (a, k, p)
    with pytest.raises(lash_amd.LashError):          # main.rs:245 "Algorithm must be either hmh, ull, or hll"
        lash_amd.params_check("minhash", 16, 0)


def test_layout_host_entries():
    """lash_layout (SURVEY App. D's unknowns as data): default, parse, sizes — host only."""
    import oracle_lib as O
    d = lash_amd.parse_layout(None)
    assert bytes(d) == bytes(O.default_layout())                      # the product's and the oracle's structs are the same 40 bytes
    assert C.sizeof(_lib.Layout) == 40 == C.sizeof(O.Layout)
    spec = "codes=ACTG,kmer=lsb,hmh_x=low,hmh_reg=be,hll_bucket=high,hmh_hdr=l,hll_hdr=pzsal,ull_hdr=pL,fastq_err=skip"
    lay = lash_amd.parse_layout(spec)
    assert bytes(lay) == bytes(O.parse_layout(spec))
    assert lash_amd.image_bytes("hmh", 0, lay) == 8 + 32768 == O.image_bytes(O.HMH, 0, O.parse_layout(spec))
    assert lash_amd.image_bytes("hll", 10, lay) == 33 + 1024 and lash_amd.image_bytes("ull", 9, lay) == 5 + 512
    assert lash_amd.header_bytes("hll") == 33 and lash_amd.header_bytes("ull") == 8 and lash_amd.header_bytes("hmh") == 0
    for bad in ("codes=ACGA", "codes=ACG", "kmer=middle", "hll_hdr=azsplazspl", "hll_hdr=x", "nonsense", "what=ever"):
        with pytest.raises(lash_amd.LashError):
            lash_amd.parse_layout(bad)
    assert _lib.load().lash_layout_check(C.byref(lay)) == lash_amd.OK
    broken = lash_amd.parse_layout(None)
    broken.base_code[2] = 0
    assert _lib.load().lash_layout_check(C.byref(broken)) == lash_amd.EINVAL
    assert lash_amd.image_bytes("hmh", 0, broken) == 0


def test_no_gpu_means_loud_failure_not_fallback():
    lib = _lib.load()
    if lib.lash_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lash_amd.LashError) as e:
        lash_amd.Context(0)
    assert e.value.code == lash_amd.ENODEV
    assert b"no CPU fallback" in lib.lash_strerror(lash_amd.ENODEV)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under lash_amd/, include/ or bench.py's GPU path may reference it,
    and the shared library must not link it."""
    for d in ("lash_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".cc")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "lash_oracle" not in txt and "oracle_lib" not in txt and "lash_or_" not in txt, os.path.join(dirpath, f)
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
