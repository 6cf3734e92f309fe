#!/usr/bin/env python3
"""tools/item_trace.py FILE — reads a LASH_ITEM_TRACE file (lash_api.hip: one line per workgroup of a direct sketch launch: when it ran on the
100 MHz wall clock and on which XCC / CU) and prints, per launch: the launch's span, how busy the chip's workgroup slots were over time
(ten deciles of the span), the per-XCC finish times, the longest items and what ran last — enough to tell a tail from a mis-balance."""
import sys
import numpy as np

launches, cur = [], None
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        cur = {"hdr": ln.strip(), "rows": []}
        launches.append(cur)
    elif cur is not None:
        cur["rows"].append([int(x) for x in ln.split()])
for li, L in enumerate(launches[-int(sys.argv[2]) if len(sys.argv) > 2 else -1:]):
    r = np.array(L["rows"], dtype=np.int64)
    ran = r[r[:, 6] > 0]
    t0, t1 = ran[:, 5].min(), ran[:, 6].max()
    span = (t1 - t0) / 100.0                                            # us
    dur = (ran[:, 6] - ran[:, 5]) / 100.0
    print(L["hdr"])
    print("launch span %.1f us; %d of %d workgroups ran a body; workgroup time: mean %.1f us, median %.1f, max %.1f; sum / span = %.1f slots busy on average"
          % (span, len(ran), len(r), dur.mean(), np.median(dur), dur.max(), dur.sum() / span))
    edges = np.linspace(t0, t1, 11)
    busy = []
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(ran[:, 6], b) - np.maximum(ran[:, 5], a), 0, None).sum()
        busy.append(ov / (b - a))
    print("slots busy per decile of the span: " + " ".join("%.0f" % x for x in busy))
    xcc = ran[:, 8] & 0xF
    print("per XCC: workgroups / busy us / last end (us after launch start): " +
          "  ".join("x%d %d / %.0f / %.0f" % (x, (xcc == x).sum(), dur[xcc == x].sum(), (ran[xcc == x][:, 6].max() - t0) / 100.0) for x in sorted(set(xcc.tolist()))))
    size = (ran[:, 4] - ran[:, 3]) * 16
    q = np.argsort(-dur)[:5]
    print("longest: " + "; ".join("item %d genome %d bytes [%d, %d) %.0f us (start +%.0f)" % (ran[i, 1], ran[i, 2], ran[i, 3] * 16, ran[i, 4] * 16, dur[i], (ran[i, 5] - t0) / 100.0) for i in q))
    q = np.argsort(-ran[:, 6])[:5]
    print("last to end: " + "; ".join("item %d genome %d bytes [%d, %d) %.0f us (start +%.0f)" % (ran[i, 1], ran[i, 2], ran[i, 3] * 16, ran[i, 4] * 16, dur[i], (ran[i, 5] - t0) / 100.0) for i in q))
    # cost per byte by the item's position inside its genome (first / middle / last third)
    for name, sel in (("first third of a genome", ran[:, 3] == 0), ("rest", ran[:, 3] != 0)):
        if sel.any():
            print("  %-24s %5d items, mean %.1f us, %.3f us per kB" % (name, sel.sum(), dur[sel].mean(), (dur[sel] / (size[sel] / 1e3)).mean()))
