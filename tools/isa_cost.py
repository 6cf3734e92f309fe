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

VOP2_CHEAP = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_lshlrev_b32",
              "v_ashrrev_i32", "v_min_u32", "v_max_u32", "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32",
              "v_subb_co_u32", "v_cndmask_b32", "v_xnor_b32", "v_min_i32", "v_max_i32"}


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
    for ln in lines:
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
        cur[-1][1].append((mn, ops, s))
    if not funcs:
        return {}
    names = list(funcs)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {d: funcs[n] for n, d in zip(names, dem)}


def block_hist(insts):
    h = collections.Counter()
    for mn, ops, _ in insts:
        h[classify(mn, ops)] += 1
    return h


VALU_CLASSES_PREFIX = ("v", )


def is_valu(cls):
    return cls.startswith("v") and cls != "vmem"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("--kernel", required=True, help="substring of the demangled kernel name")
    ap.add_argument("--costs", default=os.path.join(ROOT, "profiles", "r04", "isa_cost", "costs.json"))
    ap.add_argument("--blocks", action="store_true", help="list every basic block with its size and multiply count")
    ap.add_argument("--hot", default=None, help="comma-separated block labels to price (default: chosen automatically)")
    ap.add_argument("--kmers-per-block", type=int, default=None)
    ap.add_argument("--dump", action="store_true", help="print the instructions of the hot blocks with their classes")
    ap.add_argument("--measured", type=float, default=None, help="measured cycles per wave-k-mer, printed beside the prediction")
    args = ap.parse_args()
    ks = parse_kernels(args.listing)
    match = [n for n in ks if args.kernel in n]
    if len(match) != 1:
        sys.exit("kernel name matches %d functions: %s" % (len(match), match[:8]))
    blocks = ks[match[0]]
    print("kernel:", match[0])
    info = []
    for label, insts in blocks:
        h = block_hist(insts)
        info.append((label, len(insts), h["v_mad_u64_u32"] + h["v_mul32"], h))
    if args.blocks:
        for label, n, muls, h in info:
            if n >= 20:
                print("%-14s %5d instr  %3d multiplies  valu %4d  bfe_i32 %d" % (label, n, muls, sum(c for k, c in h.items() if is_valu(k)),
                                                                                    sum(1 for mn, _, _ in dict(blocks)[label] if mn.startswith("v_bfe_i32"))))
    costs = {}
    if os.path.exists(args.costs):
        costs = json.load(open(args.costs))["cycles"]
    if args.hot:
        hot = args.hot.split(",")
    else:
        # the unmasked per-word bodies: blocks with many multiplies and no v_bfe_i32 (the masked bodies extract 16 mask bits);
        # blocks of one straight-line word body have equal multiply counts — take the largest group
        cand = [(label, muls) for label, n, muls, h in info if muls >= 16 and not any(mn.startswith("v_bfe_i32") for mn, _, _ in dict(blocks)[label])]
        hot = [c[0] for c in cand]
    bd = dict(blocks)
    total = collections.Counter()
    for label in hot:
        total += block_hist(bd[label])
    print("hot blocks:", ", ".join("%s(%d)" % (l, len(bd[l])) for l in hot))
    if args.dump:
        for label in hot:
            print("--", label)
            for mn, ops, raw in bd[label]:
                print("   %-24s %s" % (classify(mn, ops), raw))
    nk = args.kmers_per_block
    if nk is None:
        nk = 16
    print("k-mers priced over: %d" % nk)
    tot_cycles, tot_valu = 0.0, 0
    print("%-26s %7s %9s %8s %10s" % ("class", "count", "per k-mer", "cycles", "cyc/k-mer"))
    for cls, cnt in sorted(total.items(), key=lambda kv: -kv[1]):
        c = costs.get(cls)
        cyc = cnt * c / nk if c is not None else None
        if cyc is not None and (is_valu(cls)):
            tot_cycles += cyc
        if is_valu(cls):
            tot_valu += cnt
        print("%-26s %7d %9.2f %8s %10s" % (cls, cnt, cnt / nk, "%.2f" % c if c is not None else "-", "%.1f" % cyc if cyc is not None else "-"))
    print("VALU wave-instructions per k-mer: %.2f" % (tot_valu / nk))
    if costs:
        print("predicted VALU issue cycles per wave-k-mer: %.1f%s" % (tot_cycles, ("   measured: %.1f  (ratio %.2f)" % (args.measured, tot_cycles / args.measured)) if args.measured else ""))


if __name__ == "__main__":
    main()
