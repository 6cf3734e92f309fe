#!/usr/bin/env python3
"""tools/isa_cost.py — issue-cost audit of the sketch kernels' hot loops from the compiler's own listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o sketch_kernels.s lash_amd/csrc/sketch_kernels.hip
    python tools/isa_cost.py sketch_kernels.s --kernel 'sketch_kernel<0, 0, false, 0, true, false, true>' [--blocks] [--costs costs.json]

Splits the named kernel into basic blocks, classifies every instruction by the operand/encoding classes that tools/ubench_ops.hip
measures on the GPU (cycles per wave-instruction per SIMD at 4 waves per SIMD), finds the per-word hot blocks (the unrolled
16-k-mer bodies: the blocks that hold the 64-bit multiplies) and prints count x cost per class and the sum per k-mer.
The cost table is data: profiles/r04/isa_cost/costs.json, written by tools/ubench_ops on the GPU box."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# VOP2 opcodes that tools/ubench_isa measured at double rate with vector / inline / literal sources; v_min / v_max / v_lshlrev /
# v_cndmask are VOP2 too but run at the full-rate cost and are priced as "vop3"
VOP2_CHEAP = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32", "v_not_b32",
              "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_xnor_b32"}


def operand_kinds(ops):
    kinds = []
    for o in ops:
        o = o.strip()
        if not o:
            continue
        if re.match(r"^v(\d+|\[\d+:\d+\])$", o):
            kinds.append("v")
        elif re.match(r"^(s(\d+|\[\d+:\d+\])|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|ttmp\d+)$", o):
            kinds.append("s")
        elif re.match(r"^-?(0x[0-9a-fA-F]+|\d+)$", o):
            v = int(o, 0)
            kinds.append("i" if -16 <= v <= 64 else "l")         # inline constant / 32-bit literal
        elif re.match(r"^-?\d*\.\d+$", o):
            kinds.append("i")
        else:
            kinds.append("?")
    return kinds


def classify(mn, ops):
    """-> class name (a key of the cost table)."""
    base = mn
    for suf in ("_e32", "_e64", "_sdwa", "_dpp"):
        if base.endswith(suf):
            base = base[: -len(suf)]
    enc64 = mn.endswith("_e64")
    if base.startswith("s_"):
        if base in ("s_waitcnt", "s_nop", "s_barrier", "s_sleep"):
            return "s_wait/nop"
        if base.startswith("s_load") or base.startswith("s_buffer_load"):
            return "smem"
        if base.startswith("s_cbranch") or base == "s_branch":
            return "s_branch"
        return "salu"
    if base.startswith("ds_"):
        if "read" in base or "load" in base or base.startswith("ds_bpermute") or base.startswith("ds_swizzle"):
            return "ds_read"
        if "write" in base or "store" in base:
            return "ds_write"
        return "ds_atomic"
    if base.startswith("global_") or base.startswith("buffer_") or base.startswith("flat_") or base.startswith("scratch_"):
        return "vmem"
    if not base.startswith("v_"):
        return "other"
    kinds = operand_kinds(ops[1:]) if ops else []
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return "v_cmp"
    if base in ("v_readfirstlane_b32", "v_readlane_b32", "v_writelane_b32"):
        return "v_readlane"
    if base in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
        return "v_mbcnt"
    if base == "v_mad_u64_u32" or base == "v_mad_i64_i32":
        return "v_mad_u64_u32"
    if base in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32"):
        return "v_mul32"
    if base in ("v_mul_u32_u24", "v_mad_u32_u24", "v_mul_i32_i24", "v_mad_i32_i24", "v_mul_hi_u32_u24"):
        return "v_mul24"
    if base in ("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64"):
        return "v_shift64"
    if base in ("v_lshl_add_u64",):
        return "v_lshl_add_u64"
    if base in VOP2_CHEAP and not enc64 and "sdwa" not in mn and "dpp" not in mn:
        # VOP2 encoding: src0 may be VGPR / SGPR / constant, src1 a VGPR
        src = kinds[:2]
        if base == "v_mov_b32" or base == "v_not_b32":
            src = kinds[:1]
        if all(k == "v" for k in src):
            return "vop2 v,v"
        if "l" in src:
            return "vop2 literal,v"
        if "i" in src:
            return "vop2 inline,v"
        return "vop2 s,v"
    if base in VOP2_CHEAP and enc64:
        return "vop3 (simple op, e64)"
    if "dpp" in mn:
        return "vop dpp"
    if "sdwa" in mn:
        return "vop sdwa"
    return "vop3"


def parse_kernels(path):
    """-> {demangled name: [(label, [(mnemonic, operands, raw line)])]} for every function in an amdgcn .s listing."""
    lines = open(path, errors="replace").read().splitlines()
    funcs, cur, name = {}, None, None
    files, loc = {}, None                                   # .file N "dir" "name" / .loc N line col (listings made with -gline-tables-only)
    for ln in lines:
        m = re.match(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
            continue
        m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", ln)
        if m:
            name = m.group(1)
            cur = [("entry", [])]
            funcs[name] = cur
            continue
        if cur is None:
            continue
        if re.match(r"^\s*\.end_amdhsa_kernel|^\.Lfunc_end", ln):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur.append((m.group(1), []))
            continue
        s = ln.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] if o.split() else o for o in ops]              # strips modifiers such as 'bitop3:0x96', 'offset:4'
        cur[-1][1].append((mn, ops, s if loc is None else s + "\t; @%s:%d" % (files.get(loc[0], "?"), loc[1])))
    if not funcs:
        return {}
    names = list(funcs)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {d: funcs[n] for n, d in zip(names, dem)}


def block_hist(insts):
    h = collections.Counter()
    prev = None
    for mn, ops, _ in insts:
        c = classify(mn, ops)
        # a v_mov_b32 straight behind a v_mad_u64_u32 rides in the multiply's second pass (ubench_isa "mad + mov alternating")
        if c.startswith("vop2 v,v") and mn.startswith("v_mov_b32") and prev == "v_mad_u64_u32":
            c = "v_mov after v_mad_u64_u32"
        h[c] += 1
        prev = c
    return h


def signature(insts):
    """What tells the blocks of a kernel apart without naming labels (they move with every compile)."""
    sig = collections.Counter()
    for mn, ops, _ in insts:
        b = mn
        for suf in ("_e32", "_e64", "_sdwa", "_dpp"):
            if b.endswith(suf):
                b = b[: -len(suf)]
        sig["n"] += 1
        if b in ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32"):
            sig["mul"] += 1
        if b in ("v_bfe_i32", "v_bfe_u32", "v_perm_b32", "v_ffbh_u32", "v_cmp_lt_u64", "ds_min_u32", "ds_max_i32", "ds_or_b32", "ds_read_b32", "ds_write_b32", "v_alignbit_b32"):
            sig[b.replace("v_", "").replace("_b32", "").replace("_u32", "").replace("_i32", "_i32") if not b.startswith("ds_") else b] += 1
    return sig


def is_valu(cls):
    return cls.startswith("v") and cls != "vmem"


def select(blocks, expr):
    """expr: 'mul==28&bfe_i32==0' over signature() fields; optional '@first' / '@N' keeps the first (N) matches."""
    keep = None
    if "@" in expr:
        expr, k = expr.split("@")
        keep = 1 if k == "first" else int(k)
    out = []
    for label, insts in blocks:
        sig = signature(insts)
        ok = True
        for term in expr.split("&"):
            m = re.match(r"^(\w+)(==|>=|<=)(\d+)$", term.strip())
            if not m:
                sys.exit("bad selector term: " + term)
            v, op, n = sig[m.group(1)], m.group(2), int(m.group(3))
            ok = ok and ((op == "==" and v == n) or (op == ">=" and v >= n) or (op == "<=" and v <= n))
        if ok:
            out.append(label)
    return out[:keep] if keep else out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("--kernel", required=True, help="substring of the demangled kernel name")
    ap.add_argument("--costs", default=os.path.join(ROOT, "profiles", "r04", "isa_cost", "costs.json"))
    ap.add_argument("--blocks", action="store_true", help="list every basic block with its signature")
    ap.add_argument("--section", action="append", default=[], metavar="NAME|KMERS|SELECT",
                    help="price the blocks SELECT picks as one section that covers KMERS k-mers per lane (a float: 0.044 rounds per k-mer "
                         "is KMERS = 1 / 0.044); repeatable")
    ap.add_argument("--dump", action="store_true", help="print the instructions of the selected blocks with their classes")
    ap.add_argument("--mnemonics", action="store_true", help="per section: the MNEMONIC histogram (count, per k-mer, class, cycles) and, when the listing "
                    "carries .loc lines (hipcc -gline-tables-only), the source lines the section's VALU instructions come from (VERDICT r4 next #2)")
    ap.add_argument("--measured", type=float, default=None, help="measured cycles per wave-k-mer, printed beside the prediction")
    ap.add_argument("--measured-valu", type=float, default=None, help="measured SQ_INSTS_VALU per k-mer (per lane)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--mix-factor", default=None, help="JSON {factor: F}: measured / predicted of an independent stream (tools/ubench_hash's rank-half stream "
                    "priced the same way): instructions of different classes overlap part of their issue, the pure-stream costs add up ~8 %% high")
    args = ap.parse_args()
    ks = parse_kernels(args.listing)
    match = [n for n in ks if args.kernel in n]
    if len(match) != 1:
        sys.exit("kernel name matches %d functions: %s" % (len(match), match[:8]))
    blocks = ks[match[0]]
    print("kernel:", match[0])
    # register spills: a kernel at its register cap that gains two live values starts spilling per tile (seen in round 4 as + 1.7 GB
    # of HBM writes per launch of the hll k = 21 kernel, time unchanged) — the count to compare from build to build
    sc = [(label, raw) for label, insts in blocks for mn, ops, raw in insts if mn.startswith("scratch_")]
    print("scratch instructions in the whole kernel: %d (%d stores) in %d blocks" % (len(sc), sum(r.startswith("scratch_store") for _, r in sc), len({l for l, _ in sc})))
    if args.blocks:
        for label, insts in blocks:
            sig = signature(insts)
            if sig["n"] >= 20:
                print("%-14s %s" % (label, " ".join("%s=%d" % kv for kv in sorted(sig.items()))))
    costs = json.load(open(args.costs))["cycles"] if os.path.exists(args.costs) else {}
    bd = dict(blocks)
    grand_c, grand_v, out_sections = 0.0, 0.0, []
    for spec in args.section:
        name, kmers, sel = spec.split("|")
        kmers = float(kmers)
        aside = name.startswith("~")                      # "~name": printed for comparison, not part of the kernel's sum
        name = name.lstrip("~")
        labels = select(blocks, sel)
        total = collections.Counter()
        for l in labels:
            total += block_hist(bd[l])
        print("\n== %s  [%s -> %d block(s): %s; %.4g k-mers per lane]" % (name, sel, len(labels), ", ".join("%s(%d)" % (l, len(bd[l])) for l in labels), kmers))
        n_sc = sum(mn.startswith("scratch_") for l in labels for mn, ops, raw in bd[l])
        if n_sc:
            print("   (%d scratch instructions inside this section)" % n_sc)
        if args.dump:
            for l in labels:
                print("--", l)
                for mn, ops, raw in bd[l]:
                    print("   %-28s %s" % (classify(mn, ops), raw))
        print("%-30s %7s %9s %8s %10s" % ("class", "count", "per k-mer", "cycles", "cyc/k-mer"))
        sec_c, sec_v = 0.0, 0.0
        for cls, cnt in sorted(total.items(), key=lambda kv: -kv[1]):
            c = costs.get(cls)
            cyc = cnt * c / kmers if (c is not None and is_valu(cls)) else None
            if is_valu(cls):
                sec_v += cnt / kmers
                if cyc is None:
                    print("   (no cost for class %r)" % cls)
                else:
                    sec_c += cyc
            print("%-30s %7d %9.3f %8s %10s" % (cls, cnt, cnt / kmers, ("%.2f" % c) if (c is not None and is_valu(cls)) else "-", ("%.2f" % cyc) if cyc is not None else "-"))
        print("%-30s %7s %9.3f %8s %10.2f" % ("section: VALU", "", sec_v, "", sec_c))
        if args.mnemonics:
            mh, src = collections.Counter(), collections.Counter()
            for l in labels:
                prev = None
                for mn, ops, raw in bd[l]:
                    c = classify(mn, ops)
                    if c.startswith("vop2 v,v") and mn.startswith("v_mov_b32") and prev == "v_mad_u64_u32":
                        c = "v_mov after v_mad_u64_u32"
                    prev = c
                    if not is_valu(c):
                        continue
                    mh[(mn, c)] += 1
                    at = raw.rsplit("; @", 1)
                    if len(at) == 2:
                        src[at[1]] += 1
            print("   mnemonics (VALU only):  %-28s %-28s %6s %9s %9s" % ("mnemonic", "class", "count", "per k-mer", "cyc/k-mer"))
            for (mn, c), cnt in sorted(mh.items(), key=lambda kv: -kv[1] * (costs.get(kv[0][1]) or 0)):
                cc = costs.get(c)
                print("   %22s  %-28s %-28s %6d %9.3f %9s" % ("", mn, c, cnt, cnt / kmers, ("%.2f" % (cnt * cc / kmers)) if cc is not None else "-"))
            if src:
                print("   source lines (VALU instructions per k-mer; inlined code is charged to the line it was written on):")
                for at, cnt in src.most_common(24):
                    print("   %22s  %-40s %9.3f" % ("", at, cnt / kmers))
        if aside:
            print("   (an aside: not in the sum below)")
            continue
        grand_c += sec_c
        grand_v += sec_v
        out_sections.append({"name": name, "select": sel, "blocks": len(labels), "kmers_per_lane": kmers, "valu_per_kmer": sec_v, "cycles_per_kmer": sec_c})
    factor, fsrc = 1.0, None
    if args.mix_factor and os.path.exists(args.mix_factor):
        fj = json.load(open(args.mix_factor))
        factor, fsrc = fj["factor"], fj.get("source")
    raw_c = grand_c
    grand_c *= factor
    if args.section:
        print("\nVALU wave-instructions per k-mer, sections together: %.2f%s" % (grand_v, ("   measured (SQ_INSTS_VALU): %.2f" % args.measured_valu) if args.measured_valu else ""))
        print("sum of count x pure-stream cost: %.1f cycles per wave-k-mer%s" % (raw_c, ("; x mixed-stream factor %.3f (%s)" % (factor, fsrc)) if fsrc else ""))
        print("predicted VALU issue cycles per wave-k-mer: %.1f%s" % (grand_c, ("   measured: %.1f  (predicted / measured %.2f)" % (args.measured, grand_c / args.measured)) if args.measured else ""))
        if args.measured and args.measured_valu and grand_v > 0 and args.measured_valu > grand_v:
            # what the kernel issues outside the priced blocks (per word: the window's rotation and the next reverse complement; per tile:
            # the record-boundary mask, the census, load addresses, scalar spills through v_readlane / v_writelane), at the sections' mean cost
            extra = args.measured_valu - grand_v
            with_unlisted = grand_c + extra * grand_c / grand_v
            print("instructions outside the sections (measured - listed): %.2f per k-mer; at the sections' mean cost (%.2f cycles): predicted %.1f  "
                  "(predicted / measured %.2f)" % (extra, grand_c / grand_v, with_unlisted, with_unlisted / args.measured))
    if args.json:
        json.dump({"kernel": match[0], "listing": os.path.basename(args.listing), "costs": os.path.relpath(args.costs, ROOT), "sections": out_sections,
                   "valu_per_kmer": grand_v, "cycles_per_kmer": grand_c, "pure_stream_cycles_per_kmer": raw_c, "mix_factor": factor, "measured_cycles_per_kmer": args.measured,
                   "measured_valu_per_kmer": args.measured_valu,
                   "cycles_per_kmer_with_unlisted": (grand_c + (args.measured_valu - grand_v) * grand_c / grand_v) if (args.measured_valu and grand_v > 0 and args.measured_valu > grand_v) else None},
                  open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
