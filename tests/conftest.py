import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "sole(max_bytes=None): run with the persistent small-genome kernel enabled (default LASH_SOLE_MAX, or the value given)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# GPU test modules whose genomes are (mostly) small enough for the persistent small-genome kernel: every test in them runs on BOTH routes
# (see tests/routes.py).  The others use genomes above LASH_SOLE_MAX, the dist side, or pin their routes themselves.
DUAL_ROUTE_MODULES = {"test_gpu_direct", "test_gpu_parity", "test_gpu_cli", "test_gpu_rawfiles", "test_gpu_hll_corner", "test_gpu_layout",
                      "test_gpu_rare_hashes"}


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if (mod in DUAL_ROUTE_MODULES and "_sole_mode" in metafunc.fixturenames and metafunc.definition.get_closest_marker("gpu") is not None
            and metafunc.definition.get_closest_marker("sole") is None and not os.environ.get("LASH_TEST_ONE_ROUTE")):
        metafunc.parametrize("_sole_mode", ["sliced", "sole"], indirect=True)


@pytest.fixture(autouse=True)
def _sole_mode(request):
    """Round 5: by default the library sends genomes of at most LASH_SOLE_MAX bytes to the persistent small-genome kernel
    (lash_amd/csrc/sole_kernels.hip).  Round 6: the GPU tests of DUAL_ROUTE_MODULES run twice — "sliced" pins LASH_SOLE_MAX=0 (direct pass,
    junction walks, dense tiles, stream kernel, ITEM_SOLE flush: what those tests were written for), "sole" runs the library's default and
    draws the number of persistent workgroups from {default, 1, 3} by the test's name: with few workgroups several genomes follow each
    other on the same rings and table, which is where round 5's two late bugs lived.  Tests marked `sole` run with the default (or the
    limit the marker names); every other GPU test keeps LASH_SOLE_MAX=0.  The library reads both variables on every call, and the
    command-line tests' child processes inherit them."""
    if "gpu" not in request.keywords:
        yield
        return
    import zlib
    m = request.node.get_closest_marker("sole")
    mode = getattr(request, "param", None)
    old = {v: os.environ.get(v) for v in ("LASH_SOLE_MAX", "LASH_SOLE_WGS")}
    if mode == "sole" or (m is None and mode is None and os.environ.get("LASH_TEST_SOLE_EVERYWHERE")):
        os.environ.pop("LASH_SOLE_MAX", None)
        wgs = (None, "1", "3")[zlib.crc32(request.node.nodeid.encode()) % 3]
        if wgs and "LASH_SOLE_WGS" not in os.environ:
            os.environ["LASH_SOLE_WGS"] = wgs
    elif m is None:
        os.environ["LASH_SOLE_MAX"] = "0"
    elif m.args:
        os.environ["LASH_SOLE_MAX"] = str(m.args[0])
    else:
        os.environ.pop("LASH_SOLE_MAX", None)
    yield
    for v, val in old.items():
        if val is None:
            os.environ.pop(v, None)
        else:
            os.environ[v] = val
