"""Builds liblash_gfx950.so (HIP, gfx950 only) in-tree with hipcc.  `python -m lash_amd.build [--force]`."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "liblash_gfx950.so")
SOURCES = ["lash_api.hip", "lash_plan.hip", "lash_hll_replay.hip", "lash_dist_api.hip", "sketch_set.hip", "sketch_kernels.hip", "sole_kernels.hip", "pack_kernels.hip", "fastq_check.hip", "dist_kernels.hip", "pair_planes.hip", "dist_estimators.hip"]
HEADERS = ["lash_common.h", "lash_ctx.h", "lash_internal.h", "lash_kernels.h", "lash_device.h", "sketch_rules.h", "ull_estimators.h", os.path.join(ROOT, "include", "lash_gfx950.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: liblash_gfx950.so cannot be built (there is no CPU fallback)")


OBJ_DIR = os.path.join(ROOT, "build", "obj")


def _deps():
    return [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _deps()
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """One object per source (only the stale ones are recompiled, in parallel), then one link."""
    if not force and not _stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = max(os.path.getmtime(d) for d in _deps())
    compile_flags = [f for f in FLAGS if f != "-shared"] + ["-c"]
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src + ".o")
        sp = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(sp)):
            jobs.append([hipcc] + compile_flags + ["-o", obj, sp])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(len(jobs) or 1, os.cpu_count() or 1, 8)) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [os.path.join(OBJ_DIR, s + ".o") for s in SOURCES])
    return LIB


HOST_DIR = os.path.join(CSRC, "host")
HOST_SOURCES = ["main.cpp", "sketch_files.cpp", "fastx.cpp", "pgzip.cpp", "inflate_fast.cpp", "zstd_dl.cpp", "codec_dl.cpp", "json_out.cpp", "name_order.cpp", "dist_format.cpp", "dist.cpp"]
CLI = os.path.join(PKG, "bin", "lash")
HOSTLIB = os.path.join(PKG, "liblash_host.so")


def _host_stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR)] + [os.path.join(ROOT, "include", "lash_gfx950.h")]
    return any(os.path.getmtime(d) > t for d in deps) or os.path.getmtime(LIB) > t


def build_host(force=False, verbose=False):
    """The C++ host side above the C ABI: the `lash` command line (lash_amd/bin/lash) and liblash_host.so, the same
    objects without main() behind a few extern "C" hooks so that tests can drive the FASTX / JSON / zstd code."""
    build_library(force=False, verbose=verbose)
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    common = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread"]
    link = ["-L" + PKG, "-llash_gfx950", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,$ORIGIN", "-lz", "-ldl", "-lpthread"]
    if force or _host_stale(CLI, HOST_SOURCES):
        cmd = common + ["-o", CLI] + [os.path.join(HOST_DIR, s) for s in HOST_SOURCES] + link
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if force or _host_stale(HOSTLIB, HOST_SOURCES):
        srcs = [s for s in HOST_SOURCES if s != "main.cpp"] + ["host_hooks.cpp"]
        cmd = common + ["-shared", "-o", HOSTLIB] + [os.path.join(HOST_DIR, s) for s in srcs] + link
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
