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


@pytest.fixture(autouse=True)
def _sole_mode(request):
    """Round 5: by default the library sends genomes of at most LASH_SOLE_MAX bytes to the persistent small-genome kernel
    (lash_amd/csrc/sole_kernels.hip).  The GPU tests written for the sliced kernels (direct pass, junction walks, dense tiles,
    stream kernel, ITEM_SOLE flush) use small genomes on purpose, so they pin LASH_SOLE_MAX=0 and keep testing what they were
    written for; tests marked `sole` run with the default (or with the value the marker names) and cover the new path
    (tests/test_gpu_sole.py, the fuzzers' FUZZ_SOLE knob).  The library reads the variable on every call."""
    if "gpu" not in request.keywords:
        yield
        return
    m = request.node.get_closest_marker("sole")
    old = os.environ.get("LASH_SOLE_MAX")
    if m is None and os.environ.get("LASH_TEST_SOLE_EVERYWHERE"):
        # a diagnostic run of the WHOLE suite with the library's default (small genomes through the persistent kernel everywhere): tests
        # that count the sliced kernels' launches fail for that reason; an image or census mismatch would be a bug
        os.environ.pop("LASH_SOLE_MAX", None)
    elif m is None:
        os.environ["LASH_SOLE_MAX"] = "0"
    elif m.args:
        os.environ["LASH_SOLE_MAX"] = str(m.args[0])
    else:
        os.environ.pop("LASH_SOLE_MAX", None)
    yield
    if old is None:
        os.environ.pop("LASH_SOLE_MAX", None)
    else:
        os.environ["LASH_SOLE_MAX"] = old
