"""Knobs shared by the randomized GPU runners (tests/fuzz_gpu*.py).

FUZZ_SOLE pins which genomes the persistent small-genome kernel (lash_amd/csrc/sole_kernels.hip) takes in every iteration:
    FUZZ_SOLE=0      never (LASH_SOLE_MAX=0): the sliced kernels only — direct pass, junction walks, dense tiles, stream kernel
    FUZZ_SOLE=1      the library default (genomes up to 393 216 bytes)
    FUZZ_SOLE=<n>    genomes up to n bytes (small n: most batches hold both kinds and run both launches)
unset: drawn per iteration from {0, 2 000, 50 000, default}, and the kernel's workgroup shape (LASH_SOLE_THREADS) from
{default, 64, 128, 256, 512}, and the number of its workgroups (LASH_SOLE_WGS) from {default, 1, 2, 5}: with FEW workgroups a fuzz
batch of a few dozen genomes puts several genomes on every workgroup, one after the other on the same rings and table — the paths of
a collection of a million (with the default every genome of a small batch has a workgroup of its own; a stale ring pointer survived
round 5's first campaigns that way).  FUZZ_SOLE_WGS=<n> pins it.  The library reads the variables on every call; subprocesses (the CLI
runners) inherit them."""
import os


def set_sole(rng):
    pin = os.environ.get("FUZZ_SOLE")
    if pin is None:
        choice = rng.choice(["0", "2000", "50000", None, None])
    elif pin == "1":
        choice = None
    else:
        choice = pin
    if choice is None:
        os.environ.pop("LASH_SOLE_MAX", None)
    else:
        os.environ["LASH_SOLE_MAX"] = choice
    t = rng.choice([None, None, "64", "128", "256", "512"])
    if t is None:
        os.environ.pop("LASH_SOLE_THREADS", None)
    else:
        os.environ["LASH_SOLE_THREADS"] = t
    w = os.environ.get("FUZZ_SOLE_WGS") or rng.choice([None, "1", "2", "5"])
    if w is None:
        os.environ.pop("LASH_SOLE_WGS", None)
    else:
        os.environ["LASH_SOLE_WGS"] = w
    return "sole_max=%s threads=%s wgs=%s" % (choice or "default", t or "default", w or "default")
