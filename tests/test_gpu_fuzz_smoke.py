"""A smoke tier of every randomized runner (tests/fuzz_gpu*.py) inside the `-m gpu` suite, fixed seeds, a few seconds each, so
that a regression the fuzzers would find turns the driver's GPU test record red.  The long runs stay by hand
(tools/gpu_round.sh, tools/gpu_longfuzz.sh)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize("script,iters,seed,ok", [
    ("fuzz_gpu.py", 150, 5, "fuzz ok"),                 # record / direct / pack-first routes vs the oracle
    ("fuzz_gpu_raw.py", 150, 9, "raw fuzz ok"),         # device-side FASTA / FASTQ parse, malformed records injected
    ("fuzz_gpu_cli.py", 20, 11, "cli fuzz ok"),         # `lash sketch` end to end (gz / plain, batch sizes, threads)
    ("fuzz_gpu_stream.py", 12, 7, "stream fuzz ok"),    # large-file streaming: chunk cuts, N runs over the cut
    ("fuzz_gpu_dist.py", 12, 13, "dist fuzz ok"),       # `lash dist` vs the pure-Python estimators, all three sketch types
])
def test_fuzz_runner_smoke(script, iters, seed, ok):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + HERE)
    r = subprocess.run([sys.executable, os.path.join(HERE, script), str(iters), str(seed)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and ok in r.stdout, (script, r.stdout[-1500:], r.stderr[-3000:])


def test_fuzz_with_deferred_signatures_everywhere():
    """HyperMinHash batches of long work items run sketch_kernel<..., DEFER> (signature half of the hash only for the k-mers whose
    rank can still win their bucket; lash_api.hip: from 0.6 Mbp per work item).  LASH_DEFER_MIN=0 sends EVERY direct HyperMinHash
    launch of the runner down that kernel — tiny genomes, read sets, dirt, slices — against the oracle as usual.  (The full-size
    tests take the route by themselves.)"""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + HERE, LASH_DEFER_MIN="0", FUZZ_ALGO="hmh", FUZZ_SOLE="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_gpu.py"), "150", "21"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("script,iters,seed,ok,wgs", [("fuzz_gpu.py", 150, 31, "fuzz ok", "1"), ("fuzz_gpu_raw.py", 100, 32, "raw fuzz ok", "3"),
                                                      ("fuzz_gpu_cli.py", 12, 33, "cli fuzz ok", None), ("fuzz_gpu.py", 100, 34, "fuzz ok", "3")])
def test_fuzz_with_the_persistent_kernel_everywhere(script, iters, seed, ok, wgs):
    """Round 5: genomes of at most LASH_SOLE_MAX bytes run on the persistent small-genome kernel (sole_kernels.hip).  By default the
    runners draw that limit per iteration (tests/fuzz_knobs.py: 0, 2 000, 50 000 or the library's default, and the kernel's workgroup
    shape); FUZZ_SOLE=1 pins the default, so that nearly every genome of every iteration — clean, dirty, multi-record, accumulated,
    packed first, raw FASTA / FASTQ bytes, through the CLI — takes the new path, against the oracle as usual."""
    # Round 6: the number of persistent workgroups is pinned too — 1 and 3 put every genome of an iteration's batch on the same one or three
    # workgroups, one after the other on the same rings and table (round 5's two late bugs needed exactly that); None = drawn per iteration
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + HERE, FUZZ_SOLE="1")
    if wgs:
        env["FUZZ_SOLE_WGS"] = wgs
    r = subprocess.run([sys.executable, os.path.join(HERE, script), str(iters), str(seed)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and ok in r.stdout, (script, wgs, r.stdout[-1500:], r.stderr[-3000:])
