#!/usr/bin/env python3
"""tools/ref_probe/extract_hll_bias.py — write the HLL++ empirical bias tables as the text file `lash dist --hll-bias` reads.

streaming_algorithms 0.3.3 (the crate behind the reference's HyperLogLog, Cargo.lock:1834) carries the tables of the HLL++
paper's appendix (Heule, Nunkesser, Hall 2013: rawEstimateData / biasData, one row per precision 4..18) as Rust constants.
They are Monte-Carlo measurements: they cannot be re-derived, and they are not in this repository.  After `cargo install
lash-rs` the crate's source sits unpacked under ~/.cargo/registry/src/*/streaming_algorithms-0.3.3/; this script finds the two
tables in it BY SHAPE (it was written without the crate at hand, so it does not rely on file or constant names):

  * every bracketed list of float literals in any .rs file is a row; consecutive rows under one parent are a table;
  * a candidate table has 15 rows (p = 4..18); the raw-estimate table and the bias table have the same row lengths;
  * raw estimates grow along a row and their row p starts near 0.7 * 2^p; biases shrink towards 0.

usage: extract_hll_bias.py <crate dir or .rs file> <out.txt>      exit 0 = written, 3 = tables not found
"""
import os
import re
import sys

NUM = re.compile(r"^[+-]?(\d[\d_]*)?(\.[\d_]*)?([eE][+-]?\d+)?(_?f(32|64))?$")


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def parse_number(tok):
    tok = tok.strip()
    if not tok or not NUM.match(tok) or not re.search(r"\d", tok):
        return None
    tok = re.sub(r"_?f(32|64)$", "", tok).replace("_", "")
    try:
        return float(tok)
    except ValueError:
        return None


def tables_in(src):
    """-> list of tables; a table = list of rows (lists of floats) that are consecutive children of one bracket"""
    src = strip_comments(src)
    stack, tables, orphans = [], [], []   # stack entries: [start index, list of child rows]; orphans: rows without a parent bracket
    for i, ch in enumerate(src):
        if ch == "[":
            stack.append([i, []])
        elif ch == "]" and stack:
            start, rows = stack.pop()
            body = src[start + 1:i]
            if "[" not in body:                                   # a leaf: all numbers?
                toks = [t for t in body.replace("\n", " ").split(",") if t.strip()]
                vals = [parse_number(t) for t in toks]
                if len(vals) >= 6 and all(v is not None for v in vals):
                    (stack[-1][1] if stack else orphans).append(vals)
            elif rows:
                tables.append(rows)
    # one constant per precision (no enclosing bracket): every run of 15 consecutive rows is a candidate
    for i in range(0, len(orphans) - 14):
        tables.append(orphans[i:i + 15])
    return tables


def pick(tables):
    """-> (raw rows, bias rows) for p = 4..18, or None"""
    cands = [t for t in tables if len(t) == 15]
    best = None
    for a in cands:
        for b in cands:
            if a is b or [len(r) for r in a] != [len(r) for r in b]:
                continue
            rising = sum(1 for r in a for x, y in zip(r, r[1:]) if y > x) / max(1, sum(len(r) - 1 for r in a))
            starts = all(0.3 * (1 << (4 + i)) <= a[i][0] <= 1.5 * (1 << (4 + i)) for i in range(15))
            bias_small = all(abs(b[i][-1]) < 0.2 * (1 << (4 + i)) and abs(b[i][0]) <= 1.05 * a[i][0] for i in range(15))
            if rising > 0.95 and starts and bias_small:
                score = rising
                if best is None or score > best[0]:
                    best = (score, a, b)
    return None if best is None else (best[1], best[2])


def find(path):
    files = [path] if os.path.isfile(path) else [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs if f.endswith(".rs")]
    all_tables, origin = [], {}
    for f in sorted(files):
        try:
            src = open(f, encoding="utf-8", errors="replace").read()
        except OSError:
            continue
        if src.count("[") < 30:
            continue
        for t in tables_in(src):
            origin[id(t)] = f
            all_tables.append(t)
    got = pick(all_tables)
    return (got, origin.get(id(got[0]))) if got else (None, None)


def write(out, raw, bias, origin):
    with open(out, "w") as f:
        f.write("# HLL++ empirical bias tables (Heule et al. 2013, appendix), extracted by tools/ref_probe/extract_hll_bias.py\n")
        f.write("# from %s\n" % origin)
        f.write("# format: 'p <precision> <n>' then n lines '<raw estimate> <bias>'\n")
        for i in range(15):
            f.write("p %d %d\n" % (4 + i, len(raw[i])))
            for r, b in zip(raw[i], bias[i]):
                f.write("%r %r\n" % (r, b))


def main(argv):
    if len(argv) != 3:
        print(__doc__, file=sys.stderr)
        return 2
    got, origin = find(argv[1])
    if not got:
        print("no pair of 15-row float tables shaped like rawEstimateData / biasData under %s" % argv[1], file=sys.stderr)
        return 3
    write(argv[2], got[0], got[1], origin)
    print("wrote %s (rows of %s samples) from %s" % (argv[2], ", ".join(str(len(r)) for r in got[0]), origin))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
