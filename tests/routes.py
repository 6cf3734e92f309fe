"""Which kernel family a GPU test's genomes take, and route-aware forms of the launch-counter assertions (round 6).

Since round 5 the library sends genomes of at most LASH_SOLE_MAX bytes (default 393 216) to the persistent small-genome kernel
(lash_amd/csrc/sole_kernels.hip) and the others to the sliced kernels (sketch_kernel<DIRECT> / stream_sketch_kernel).  tests/conftest.py
runs every GPU test of the modules that use small genomes TWICE — "sliced" (LASH_SOLE_MAX=0: the kernels the tests were written for) and
"sole" (the library default: what a user of the shipped library gets, utils.rs:450-509 — one sketch per file whatever its size) — so the
assertions on lash_timing's launch counters say what must hold on each route."""
import os


def sole_on():
    """True when the library may take small genomes to the persistent kernel in this test (the variable is read on every call)."""
    return os.environ.get("LASH_SOLE_MAX") != "0"


def assert_ascii_route(tm, direct=1):
    """The batch was sketched from its ASCII bytes in one go: `direct` launches of the sliced direct kernel — or, with the persistent
    kernel on, at most that many plus the persistent launches that took the small genomes (a batch of nothing but small genomes has no
    sliced launch at all)."""
    if sole_on():
        assert tm["direct_launches"] <= direct and tm["direct_launches"] + tm["sole_launches"] >= 1, tm
    else:
        assert tm["direct_launches"] == direct and tm["sole_launches"] == 0, tm
