#!/usr/bin/env python3
"""tools/box_info.py — what the host side of the GPU box really offers (CPU baseline provenance)."""
import os
import shutil
import subprocess


def read(path):
    try:
        return open(path).read().strip()
    except Exception as e:
        return "<%s>" % e.__class__.__name__


def main():
    print("os.cpu_count()            ", os.cpu_count())
    print("len(sched_getaffinity(0)) ", len(os.sched_getaffinity(0)))
    print("/sys/fs/cgroup/cpu.max    ", read("/sys/fs/cgroup/cpu.max"))
    print("cfs_quota_us / period_us  ", read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"))
    print("cpuset.cpus.effective     ", read("/sys/fs/cgroup/cpuset.cpus.effective"))
    model = [l.split(":", 1)[1].strip() for l in read("/proc/cpuinfo").splitlines() if l.startswith("model name")]
    print("cpu model                 ", model[0] if model else "?", "x", len(model))
    print("MemTotal                  ", [l for l in read("/proc/meminfo").splitlines() if l.startswith("MemTotal")])
    for tool in ("gcc", "g++", "hipcc", "cargo", "rustc", "zstd", "gzip", "pigz"):
        print("which %-20s" % tool, shutil.which(tool))
    try:
        print(subprocess.check_output(["lscpu"], text=True, timeout=10))
    except Exception as e:
        print("lscpu:", e)


if __name__ == "__main__":
    main()
