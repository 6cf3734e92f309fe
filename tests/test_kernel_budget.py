"""The judged sketch kernels sit at their 128-register cap: two more live values across the tile loop and hipcc spills per tile — which
shows as HBM writes (round 4: + 1.7 GB per launch of the hll k = 21 kernel at unchanged time), not as a failing test.  This one compiles
sketch_kernels.hip to a listing (about two minutes, once per source state: the listing is cached under /tmp) and holds the scratch
instruction counts of those kernels to what the committed profiles were measured with (profiles/r04/isa_cost/*.txt, line 2)."""
import hashlib
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lash_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))

BUDGET = {  # kernel (demangled prefix) -> most scratch instructions it may hold
    "void lash::sketch_kernel<0, 0, false, 0, true, true>": 20,    # hmh k=16, direct, deferring (the default bench): 18
    "void lash::sketch_kernel<1, 2, false, 0, true, false>": 17,   # hll k>16 (configs[2]): 15
    "void lash::sketch_kernel<2, 0, false, 0, true, false>": 11,   # ull k=16 (the reads shape): 9 (round 6, with the XCD skew of the item index: 11)
    # round 6: the rule variants are the same kernels with another hash half / bucket side — and must stay inside the same budget
    "void lash::sketch_kernel<0, 0, true, 0, true, true>": 20,     # hmh x = low half, direct, deferring: 18
    "void lash::sketch_kernel<1, 2, true, 0, true, false>": 17,    # hll bucket = top bits, k>16: 15
}


def _listing():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hip")) and (f.startswith("sketch_kernels") or f.endswith(".h")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    out = "/tmp/lash_sketch_kernels_%s.s" % h.hexdigest()[:16]
    if not os.path.exists(out):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-S", "--cuda-device-only",
                               "-o", out + ".tmp", os.path.join(CSRC, "sketch_kernels.hip")], stderr=subprocess.DEVNULL)
        os.replace(out + ".tmp", out)
    return out


def test_judged_kernels_keep_their_scratch_budget():
    import isa_cost
    ks = isa_cost.parse_kernels(_listing())
    for prefix, limit in BUDGET.items():
        match = [n for n in ks if n.startswith(prefix)]
        assert len(match) == 1, (prefix, match)
        n = sum(mn.startswith("scratch_") for _, insts in ks[match[0]] for mn, _, _ in insts)
        assert n <= limit, "%s: %d scratch instructions (budget %d): a new live value across the tile loop? see tools/isa_audit.sh" % (prefix, n, limit)
