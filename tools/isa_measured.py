#!/usr/bin/env python3
"""tools/isa_measured.py <pmc_summary.txt> <kernel substring> <k-mers per launch> -> "--measured C --measured-valu V" for tools/isa_cost.py:
C = cycles per wave-k-mer per SIMD = launch time x shader clock / (k-mers / (256 CUs x 4 SIMDs x 64 lanes)), the clock from
GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / launch time of the same counter pass; V = SQ_INSTS_VALU / (k-mers / 64)."""
import re
import sys


def main():
    path, sub, kmers = sys.argv[1], sys.argv[2], float(sys.argv[3])
    cur, vals, ns = None, {}, {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(.+?)\s+dispatches=(\d+) avg_ns=(\d+)", line)
        if m:
            cur = (m.group(1), m.group(2)) if sub in m.group(2) else None
            if cur:
                ns[cur[0]] = float(m.group(4))
            continue
        m = re.match(r"\s+(\S+)\s+avg/dispatch = (\S+)", line)
        if m and cur:
            vals[m.group(1)] = (float(m.group(2)), cur[0])
    gui, p_gui = vals["GRBM_GUI_ACTIVE"]
    clock = gui / 8.0 / (ns[p_gui] * 1e-9)
    insts, p_i = vals["SQ_INSTS_VALU"]
    t = ns[p_i] * 1e-9
    cyc = t * clock / (kmers / (256 * 4 * 64))
    print("--measured %.1f --measured-valu %.2f" % (cyc, insts / (kmers / 64.0)))
    sys.stderr.write("clock %.3f GHz (GRBM_GUI_ACTIVE), launch %.3f ms under the counter pass\n" % (clock / 1e9, t * 1e3))


if __name__ == "__main__":
    main()
