#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: average counter value per dispatch, per kernel.
usage: tools/pmc_summary.py gpurun_out/prof_<tag> [kernel-substring ...]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    filters = sys.argv[2:]
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv"))):
        run = f.split(os.sep)[-3]
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = name.split("(")[0].replace("void ", "")
            if filters and not any(s in short for s in filters):
                continue
            rows[(run, short)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[(run, short)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (run, short), ctrs in sorted(rows.items()):
        d = dur[(run, short)]
        print("%-22s %-50s dispatches=%d avg_ns=%.0f" % (run, short[:50], len(d), sum(d) / len(d)))
        for c, v in sorted(ctrs.items()):
            print("      %-28s avg/dispatch = %.6g" % (c, sum(v) / len(v)))


if __name__ == "__main__":
    main()
