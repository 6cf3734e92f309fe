#!/usr/bin/env python3
"""bench.py — k-mers/s sketched on MI355X, BASELINE.json's metric.

One step = one pass of the hot path (pack -> sketch -> finalize, i.e. the body of the reference's
files.par_iter().map(...) at utils.rs:450-509) over one batch of synthetic genomes whose ASCII records are already
resident in HBM; the images `S::save` would write stay in HBM.  Default workload: 12 500 synthetic 5 Mbp genomes per
GPU (62.5 GB resident; BASELINE configs[3]'s per-GPU share, and the north-star's ">= 10 000 genomes"), -a hmh -k 16,
seed 42 (SURVEY.md §8(d) generator); `--genomes 1000` is configs[1].  Genomes shard across ranks with no data-path
collective (weak scaling: every rank sketches its own 12 500 genomes).

    python bench.py                       # 1 GPU
    python bench.py --gpus N              # N > 1 without a rendezvous in the environment: starts the line below as a child process
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement" for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_genome(L, image_bytes, ascii_input):
    """SURVEY.md §8(d), per genome of L surviving bases with an S-byte image: a kernel whose input is the ASCII
    records moves L + S bytes; one whose input is the packed 2-bit form moves ceil(L/4) + S."""
    return (L if ascii_input else (L + 3) // 4) + image_bytes


def host_provenance():
    """What the host side of this box really offers: logical CPUs, the affinity mask, the cgroup CPU
    quota — the effective core count is the minimum of the three — and the CPU model."""
    ncpu = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = ncpu
    quota, quota_src = None, None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                     # cgroup v2
        quota_src = "cpu.max=%s/%s" % (q, per)
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota_src = "cfs_quota_us=%d/cfs_period_us=%d" % (q, per)
            if q > 0:
                quota = q / per
        except Exception:
            pass
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    eff = min(ncpu, aff, int(quota + 0.5) if quota else ncpu)
    return {"logical_cpus": ncpu, "affinity": aff, "cgroup_quota_cpus": quota, "cgroup": quota_src, "effective_cores": max(1, eff),
            "cpu_model": model}


XGMI_PEAK_GBS = 7 * 153.0       # the same guide: 7 xGMI links x ~153 GB/s per GPU, point to point


def device_identity(torch, local_rank):
    """What tells one GPU from another in a JSON line: uuid / PCI bus id from the device properties (no extra HIP state), the shader
    clock from sysfs when the node shows it.  Every field is best effort; `id` is the one the distinct-device count is taken on."""
    d = {"local_rank": local_rank}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        d["name"] = pr.name
        d["arch"] = getattr(pr, "gcnArchName", None)
        u = getattr(pr, "uuid", None)
        d["uuid"] = str(u) if u is not None else None
        bus = [getattr(pr, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        d["pci"] = "%04x:%02x:%02x" % tuple(bus) if all(b is not None for b in bus) else None
        d["cus"] = getattr(pr, "multi_processor_count", None)
    except Exception as e:                                 # (a gloo dry run on a CPU-only host lands here)
        d["error"] = repr(e)
    d["visible"] = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    d["id"] = d.get("uuid") if d.get("uuid") and set(d["uuid"]) - set("0-") else (d.get("pci") or "local_rank %d" % local_rank)
    try:
        import glob
        for card in glob.glob("/sys/class/drm/card*/device"):
            if d.get("pci") and os.path.basename(os.path.realpath(card)).lower().startswith(d["pci"]):
                cur = [ln for ln in open(os.path.join(card, "pp_dpm_sclk")).read().splitlines() if ln.rstrip().endswith("*")]
                d["sclk"] = cur[0].split(":", 1)[1].strip(" *") if cur else None
    except Exception:
        pass
    return d


def rank_evidence(torch, dist, dev, rank, world, local_rank, fields):
    """evidence, in the ONE JSON line, that the collective library saw N ranks on N distinct devices — and what
    each of them measured.  Every rank contributes `fields` (its own wall clock over the timed steps, its kernels' HIP-event time, ...)
    plus host name, process id and device identity through one all_gather_object, and a 1 through an all_reduce(SUM) ON THE BACKEND'S
    OWN DEVICE TENSORS (nccl = RCCL: the sum is the number of ranks the communicator really joined).  Outside the timed region.
    The reference's counterpart is rayon's `files.par_iter()` (utils.rs:450-452) / `par_iter` over reference rows (utils.rs:150)."""
    import socket
    mine = dict(fields, rank=rank, hostname=socket.gethostname(), pid=os.getpid(), device=device_identity(torch, local_rank))
    if not dist:
        per = [mine]
        seen, backend = 1, None
    else:
        per = [None] * world
        dist.all_gather_object(per, mine)
        backend = dist.get_backend()
        one = torch.ones(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        seen = int(one.item())
    per.sort(key=lambda r: r["rank"])
    ids = [r["device"]["id"] + "@" + r["hostname"] for r in per]
    el = [r["elapsed_ms"] for r in per]
    return {"backend": backend, "ranks_seen": seen, "ranks_expected": world, "devices": ids, "devices_distinct": len(set(ids)),
            "per_rank_ms": el, "imbalance": (max(el) / min(el)) if min(el) > 0 else None,
            "per_rank": per}


def native_oracle():
    """the CPU leg is timed on an oracle built `-O3 -march=native` for THIS host when gcc is here (temporary
    directory; the shipped generic .so otherwise).  Returns (path or None, flags string)."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if not gcc:
        return None, "shipped oracle/liblash_oracle.so (-O3, baseline x86-64; no gcc on this box)"
    d = tempfile.mkdtemp(prefix="lash_oracle_native_")
    so = os.path.join(d, "liblash_oracle_native.so")
    flags = ["-O3", "-march=native", "-std=c11", "-fPIC", "-shared"]
    try:
        subprocess.check_call([gcc] + flags + ["-o", so, os.path.join(ROOT, "oracle", "lash_oracle.c"), "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        return so, "rebuilt on this host: gcc " + " ".join(flags[:2])
    except Exception:
        return None, "shipped oracle/liblash_oracle.so (-O3, baseline x86-64; the native rebuild failed)"


def cpu_baseline(algo, k, p, seed, L, target_s, first_genome, check_images, layout=None):
    """The only place bench.py touches oracle/.  Times the CPU oracle (a port: the Rust reference cannot be built here) with
    the reference's parallel structure — one task per genome, dynamic scheduling (utils.rs:450-452) — on a bounded sample of
    the same synthetic workload: (i) parse-inclusive, from FASTA text in memory (80-column lines; needletail-like parse +
    the per-file closure, utils.rs:452-508), (ii) sketch-only, from pre-loaded sequences.  Threads = the EFFECTIVE cores
    (affinity and cgroup quota honoured, see host_provenance()); a short scan around that number is reported.
    `check_images` {genome index: GPU image}: the oracle also sketches those genomes of the GPU workload and the images must
    be identical (returned as the flag)."""
    prov = host_provenance()
    so, flags = native_oracle()
    if so:
        os.environ["LASH_ORACLE_LIB"] = so
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    eff = prov["effective_cores"]
    n = max(8, min(2 * eff, 256))
    gen0 = 10_000_000                     # genome ids disjoint from the GPU workload
    seqs = np.concatenate([O.synth_genome(gen0 + g, L) for g in range(n)])
    rec_off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    goff = np.arange(n + 1, dtype=np.uint64)
    algo_id = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}[algo]
    lay = O.parse_layout(layout) if layout else None          # (--layout: the oracle restates the same alternative rule)

    def run(threads, genomes):
        done, el = 0, 0.0
        while el < 1.0:                                       # >= 1 s per point: a cgroup quota is enforced per 100 ms period
            t0 = time.perf_counter()
            O.sketch_genomes(algo_id, k, p, seed, seqs[:genomes * L], rec_off[:genomes + 1], goff[:genomes + 1], threads=threads, layout=lay)
            el += time.perf_counter() - t0
            done += genomes
        return done * (L - k + 1) / el

    # threads = the effective core count (the scan over 0.5x..4x of it bought 9 % on a 16-CPU cgroup and took longer
    # than the GPU phase of the whole run); LASH_BENCH_CPU_SCAN=1 brings the scan back
    cands = [eff]
    if os.environ.get("LASH_BENCH_CPU_SCAN"):
        cands = sorted({eff, max(1, eff // 2), min(prov["affinity"], 2 * eff), min(prov["affinity"], 4 * eff)}, reverse=True)
    scan = {}
    if len(cands) > 1:
        for T in cands:
            scan[T] = run(T, min(n, max(2 * T, 8)))
        best = max(scan, key=scan.get)
    else:
        best = eff
    done, elapsed = 0, 0.0
    while elapsed < target_s:
        t0 = time.perf_counter()
        O.sketch_genomes(algo_id, k, p, seed, seqs, rec_off, goff, threads=best, layout=lay)
        elapsed += time.perf_counter() - t0
        done += n
    kmers = done * (L - k + 1)
    # (i) parse-inclusive: the same genomes as 80-column FASTA text
    nf = min(n, max(4, best))
    files = []
    for g in range(nf):
        body = seqs[g * L:(g + 1) * L]
        full = body[: (L // 80) * 80].reshape(-1, 80)
        txt = np.concatenate([full, np.full((full.shape[0], 1), 10, np.uint8)], axis=1).reshape(-1)
        files.append(b">g%d\n" % g + txt.tobytes() + body[(L // 80) * 80:].tobytes() + b"\n")
    pdone, pelapsed = 0, 0.0
    while pelapsed < max(1.0, target_s / 3):                 # long enough that a cgroup CPU quota cannot be out-run by a burst
        t0 = time.perf_counter()
        O.sketch_files(algo_id, k, p, seed, files, threads=best, layout=lay)
        pelapsed += time.perf_counter() - t0
        pdone += nf
    parse_rate = pdone * (L - k + 1) / pelapsed
    ok = True
    for g, got in check_images.items():
        host = O.synth_genome(first_genome + g, L)
        want = O.sketch_genomes(algo_id, k, p, seed, host, np.array([0, L], np.uint64), np.array([0, 1], np.uint64), layout=lay)[0]
        ok = ok and bool(np.array_equal(got, want))
    return {"value": kmers / elapsed, "unit": "k-mers/s", "cores": best, "kind": "port",
            "sample": "%d synthetic %d-bp genomes, %s k=%d, in-memory sequences (no FASTA parse), %.1f s on the wall clock = %.0f core-seconds of CPU work; "
                      "oracle/lash_oracle.c (%s), one task per genome over %d threads (effective cores %d: %d logical, affinity %d, "
                      "cgroup %s%s)"
                      % (done, L, algo, k, elapsed, elapsed * best, flags, best, eff, prov["logical_cpus"], prov["affinity"], prov["cgroup"],
                         ("; scan over %s threads" % sorted(scan)) if scan else ""),
            "parse_inclusive_value": parse_rate,
            "parse_inclusive_sample": "%d passes over %d of those genomes as 80-column FASTA text in memory (%.1f s): needletail-like parse + "
                                      "filter + 2-bit copy + k-mers + sketch per file (utils.rs:452-508), %d threads"
                                      % (pdone // nf, nf, pelapsed, best),
            "host": prov, "oracle_build": flags,
            "thread_scan": {str(T): v for T, v in sorted(scan.items())}}, ok


def valu_roofline(kmers_per_launch, sketch_ms, direct, algo, k, run_ubench=True, defer=False, p=0, reads=False, sole=False):
    """The binding roofline of the sketch kernel is integer-VALU issue, not HBM.  Its ceiling is
    MEASURED: tools/ubench_hash runs the kernel's per-k-mer instruction stream (window, reverse complement, xxh3_128, register
    rule, LDS atomic) from registers, no HBM, at the kernel's occupancy; run here when the binary is built, else the
    committed round-2 figure.  insts_per_kmer comes from the committed SQ_INSTS_VALU counter pass (profiles/valu.json)."""
    import re
    import subprocess
    floor, src, clock_ghz = None, None, None
    exe = os.path.join(ROOT, "tools", "ubench_hash")
    line = {"hmh": r"\+ ds_max_u32", "hll": "hll p14 k21 stream", "ull": "ull p12 k16 stream"}[algo]
    if defer:
        line = "defer: hmh k16 stream"                   # sketch_kernel<DIRECT, DEFER>'s own stream: rank half, read back, test, append, drain
    if run_ubench and os.path.exists(exe):
        try:
            out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
            m = re.search(line + r".*\(([0-9.e+]+) k-mers/s chip-wide\)", out)
            mc = re.search(line + r".*clock ([0-9.]+) GHz", out)
            if mc:
                clock_ghz = float(mc.group(1))
            if m:
                floor, src = float(m.group(1)), ("tools/ubench_hash run in this process' job (%s instruction stream from registers%s)"
                                                  % (line.replace("\\", "").replace("+ ds_max_u32", "hmh k=16"),
                                                     "" if (algo, k) in (("hmh", 16), ("hll", 21), ("ull", 16)) else "; measured for that k, not this one"))
        except Exception:
            pass
    per, vsrc = None, None
    vpath = os.path.join(ROOT, "profiles", "valu.json")
    if os.path.exists(vpath):
        try:
            vj = json.load(open(vpath))
            key = "%s%s_k%d%s" % ("direct_" if direct else "", algo, k, "_defer" if defer else "")
            if floor is None and "issue_floor_kmers_per_s" in vj.get(key, {}):
                floor, src = vj[key]["issue_floor_kmers_per_s"], vj[key].get("floor_source", "profiles/valu.json")
            per, vsrc = vj.get(key, {}).get("valu_insts_per_kmer"), vj.get(key, {}).get("source")
        except Exception:
            pass
    defer_note = None
    if defer:
        defer_note = ("this launch ran sketch_kernel<DIRECT, DEFER> (signature half of xxh3_128 only for the k-mers whose rank can still win "
                      "their bucket): the ceiling is that kernel's own stream in tools/ubench_hash ('defer: hmh k16 stream'), the table "
                      "filling as in a 5 Mbp work item")
    rate = kmers_per_launch / (sketch_ms * 1e-3) if sketch_ms > 0 else 0.0
    # The instruction-mix-weighted ceiling: tools/isa_audit.sh prices the launched kernel's hot blocks from the
    # compiler's listing with the per-class issue costs tools/ubench_isa measured on the GPU (profiles/r04/isa_cost/costs.json):
    # sum over classes of count x cost = VALU issue cycles per wave-k-mer; the chip has CUs x 4 SIMDs of them per cycle.
    mix = None
    # each audit prices ONE kernel variant on ONE shape: it applies only to the launch that matches it (ADVICE r4 — the ull audit is of the
    # p = 12 reads-shaped launch, the hll one of the p = 14 ds_max kernel, none of them of the persistent small-genome kernel)
    audit = {("hmh", True, 0, False): "hmh_k16_defer", ("hll", False, 14, False): "hll_p14_k21",
             ("ull", False, 12, True): "ull_p12_k16_reads"}.get((algo, bool(defer), 0 if algo == "hmh" else p, bool(reads)))
    if sole:
        audit = None
    apath, around = None, "r05"
    for around in ("r05", "r04"):                     # this round's audit (the HyperLogLog selector fixed: profiles/r05/floor_hll_ull.md), else round 4's
        apath = os.path.join(ROOT, "profiles", around, "isa_cost", "%s.json" % audit) if audit else None
        if apath and os.path.exists(apath):
            break
    if direct and apath and os.path.exists(apath) and (algo, k) in (("hmh", 16), ("hll", 21), ("ull", 16)):
        try:
            aj = json.load(open(apath))
            ghz = clock_ghz or 2.4
            ceil_rate = 256 * 4 * ghz * 1e9 / aj["cycles_per_kmer"] * 64.0
            mix = {"value": ceil_rate, "unit": "k-mers/s", "frac": rate / ceil_rate if ceil_rate else None,
                   "cycles_per_wave_kmer": aj["cycles_per_kmer"], "valu_per_kmer_in_listing": aj["valu_per_kmer"], "clock_ghz": ghz,
                   # the same with the instructions the kernel issues OUTSIDE the priced blocks (measured SQ_INSTS_VALU - listed) at the blocks' mean cost
                   "cycles_per_wave_kmer_with_unlisted": aj.get("cycles_per_kmer_with_unlisted"),
                   "frac_with_unlisted": (rate / (256 * 4 * ghz * 1e9 / aj["cycles_per_kmer_with_unlisted"] * 64.0)) if aj.get("cycles_per_kmer_with_unlisted") else None,
                   "clock_source": "tools/ubench_hash (s_memtime / s_memrealtime) in this job" if clock_ghz else "datasheet maximum",
                   "sections": [{"name": x["name"], "valu_per_kmer": x["valu_per_kmer"], "cycles_per_kmer": x["cycles_per_kmer"]} for x in aj["sections"]],
                   "source": "profiles/%s/isa_cost/%s.txt (tools/isa_audit.sh: hot blocks of %s from hipcc's listing x the issue costs of "
                             "profiles/r04/isa_cost/costs.json, measured by tools/ubench_isa at 4 waves per SIMD, x the mixed-stream factor %.3f of "
                             "profiles/r04/isa_cost/mix_factor.json: tools/ubench_hash's rank-half stream measured / priced the same way); per-word "
                             "and per-tile bookkeeping outside the priced blocks is not in the sum"
                             % (around, audit, aj["kernel"].replace("void lash::", "").split("(")[0], aj.get("mix_factor", 1.0))}
        except Exception:
            mix = None
    # the absolute figure beside the self-referential one: wave-instructions issued per second against the chip's issue peak
    # (256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles: MI355X_MICROARCH.md, Execution model) at the clock the
    # ubench run measured in this job (else the 2.4 GHz maximum).  insts_per_kmer is a per-LANE count: SQ_INSTS_VALU / (k-mers / 64).
    absolute = None
    if per:
        wave_instr_per_s = rate * per / 64.0
        peak_issue = 256 * 4 * (clock_ghz or 2.4) * 1e9 / 2.0
        absolute = {"wave_instr_per_s": wave_instr_per_s, "peak_wave_instr_per_s": peak_issue, "frac": wave_instr_per_s / peak_issue,
                    "clock_ghz": clock_ghz or 2.4, "clock_source": "tools/ubench_hash (s_memtime / s_memrealtime) in this job" if clock_ghz else "datasheet maximum",
                    "note": "the remainder is 4-cycle instructions (multiplies, alignbit, 3-operand VOP3) issued at half rate, not idle slots"}
    return {"bound": "valu-issue", "achieved": rate, "peak": floor, "unit": "k-mers/s", "frac": (rate / floor) if floor else None,
            "mix_ceiling": mix, "insts_per_kmer": per, "insts_source": (vsrc + " — from the committed counter pass, not measured in this run") if vsrc else None,
            "peak_source": src, "absolute_issue": absolute, "deferred_signatures": defer_note,
            "note": "mix_ceiling = chip issue slots / (sum over instruction classes of count x measured issue cost) for the launched kernel "
                    "variant; peak / frac = the same stream run from registers by tools/ubench_hash (self-referential, kept as the secondary "
                    "figure); absolute_issue = issued wave-instructions against one per 2 cycles (no VALU mix reaches that: "
                    "profiles/r04/isa_cost/ubench_isa_w4.txt)"}


def allpairs_bench(args, ctx, torch, dist, dev, rank, world, algo, k, p, seed, L, G, d_seq, d_rec, goff, rec_off, d_img, ib):
    """BASELINE configs[3] in the shape one node allows: every rank sketches its G genomes (shard), ONE collective — the
    all-gather of the finished images over RCCL — then every rank holds all N sketches as a resident set (lash_sketch_set:
    operands of the pair kernels built once per step) and computes the pair statistics of its two bands of the lower triangle
    (rows handed out so that every rank owns the same number of printed pairs, lash_amd.allpairs.row_bands), tiles above the
    diagonal skipped.  A step is timed by HIP-synchronised wall clock like the sketch bench; the O(pairs) host arithmetic and
    text (liblash_host.so) are timed beside it on a bounded sample of this rank's rows."""
    import numpy as np
    import lash_amd
    from lash_amd.allpairs import row_bands
    N = G * world
    every = torch.empty((N, ib), dtype=torch.uint8, device=dev)
    bands = row_bands(N, rank, world)
    cap = 1 << 27                                          # pair-table entries per call
    blocks = []
    for b_lo, b_hi in bands:
        b0 = b_lo
        while b0 < b_hi:
            b1 = min(b_hi, b0 + max(1, cap // (b0 + 1)))
            while b1 > b0 + 1 and (b1 - b0) * b1 > cap:
                b1 -= 1
            blocks.append((b0, b1))
            b0 = b1
    big = max((b1 - b0) * b1 for b0, b1 in blocks)
    o_c = torch.empty(big, dtype=torch.int32, device=dev)
    o_n = torch.empty(big, dtype=torch.int32, device=dev)
    o_u = torch.empty(big, dtype=torch.float64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    # one explicit stream for everything: the library's kernels, the collective (RCCL orders itself against the current stream)
    # and the events — torch's default stream is the NULL stream, which lash_ctx_set_stream takes as "use your own"
    side = torch.cuda.Stream(device=dev)
    ctx.set_stream(side)
    torch.cuda.set_stream(side)
    lib = lash_amd.load()

    def step(timed=False):
        if timed:
            ev[0].record()
        ctx.sketch_batch_device(algo, k, p, seed, d_seq, d_rec, G, goff, rec_off, d_img)
        if timed:
            ev[1].record()
        if dist and dist.get_backend() != "nccl":
            # LASH_BENCH_BACKEND=gloo, the launch dry run (ranks share devices): the same collective on host tensors
            mine, gathered = d_img.cpu(), torch.empty(every.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(gathered, mine)
            every.view(-1).copy_(gathered)
        elif dist:
            dist.all_gather_into_tensor(every.view(-1), d_img)
        else:
            every.view(-1).copy_(d_img)
        if timed:
            ev[2].record()
        s = ctx.sketch_set(algo, p, every)
        s.prepare()
        for b0, b1 in blocks:
            ctx._check(lib.lash_sketch_set_pair_block_device(ctx._h, s._h, b0, b1, s._h, b1, 1, 0, o_c.data_ptr(), o_n.data_ptr(), o_u.data_ptr()))
        if timed:
            ev[3].record()
        return s

    for _ in range(2 + args.warmup):
        step().free()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step().free()                                     # (free synchronizes the stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    s = step(timed=True)
    torch.cuda.synchronize()
    stage = {"sketch": ev[0].elapsed_time(ev[1]), "gather": ev[1].elapsed_time(ev[2]), "set + pairs": ev[2].elapsed_time(ev[3])}
    # the collective alone, on its own: the stage bracket above ends when the gather's kernel has run on THIS rank, which includes waiting
    # for the slowest rank's sketches; here every rank has its images ready and enters together (barrier), five gathers, the mean
    gather_ms = None
    if dist and dist.get_backend() == "nccl":
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        torch.cuda.synchronize()
        g0.record()
        for _ in range(5):
            dist.all_gather_into_tensor(every.view(-1), d_img)
        g1.record()
        torch.cuda.synchronize()
        gather_ms = g0.elapsed_time(g1) / 5
    recv_bytes = (world - 1) * G * ib                      # what one rank receives per gather (it already holds its own share)
    evidence = rank_evidence(torch, dist, dev, rank, world, int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1),
                             {"elapsed_ms": elapsed / args.steps * 1e3, "sketch_ms": stage["sketch"], "gather_stage_ms": stage["gather"],
                              "pairs_ms": stage["set + pairs"], "gather_ms": gather_ms,
                              "gather_GBps": (recv_bytes / (gather_ms * 1e-3) / 1e9) if gather_ms else None,
                              "rows": sum(b - a for a, b in bands), "first_genome": int(rank * G)})
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gms = [r["gather_ms"] for r in evidence["per_rank"] if r.get("gather_ms")]
    evidence["gather"] = {"bytes_received_per_rank": recv_bytes, "gather_ms_max": max(gms) if gms else None,
                          "gather_GBps_per_rank_min": (recv_bytes / (max(gms) * 1e-3) / 1e9) if gms else None,
                          "xgmi_peak_GBps_per_gpu": XGMI_PEAK_GBS,
                          "frac_of_xgmi": (recv_bytes / (max(gms) * 1e-3) / 1e9 / XGMI_PEAK_GBS) if gms else None,
                          "note": "one all_gather_into_tensor of the finished images (lash_amd/allpairs.py does the same); gather_ms = the collective alone, ranks entering "
                                  "together, mean of 5; a ring all-gather over point-to-point xGMI is bound by ONE link (~153 GB/s), a direct one by all seven"}
    # host side on a bounded sample of this rank's rows: cardinalities (GPU histograms + host finish) and rows -> text
    t1 = time.perf_counter()
    card = s.cardinalities()
    card_s = time.perf_counter() - t1
    from lash_amd.allpairs import _Formatter
    from lash_amd.shard import effective_cores
    names = ["g%06d.fa" % i for i in range(N)]
    fmt = _Formatter(names, card)
    b0 = blocks[-1][0]
    b1 = min(blocks[-1][1], b0 + 64)
    st = s.pair_block(b0, b1, n_cols=b1, triangle=True)
    threads = max(1, 2 * effective_cores() // world)
    fd = os.open(os.devnull, os.O_WRONLY)
    t1 = time.perf_counter()
    fmt.block(lash_amd.ALGOS[algo], p, k, 1, False, None, b0, b1, st, b1, False, threads, fd)
    host_pairs_per_s = sum(range(b0 + 1, b1 + 1)) / (time.perf_counter() - t1)
    os.close(fd)
    fmt.close()
    s.free()
    if rank == 0:
        kmers = G * (L - k + 1) * world * args.steps
        pairs = N * (N + 1) // 2 * args.steps             # all ranks' bands together: the printed triangle per step
        print(json.dumps({
            "metric": "k-mers/s sketched + all-vs-all (%s, k=%d)" % (algo, k), "value": kmers / elapsed, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic" if not dist or dist.get_backend() == "nccl" else "synthetic (LAUNCH DRY RUN: ranks share GPUs over gloo, not a scaling measurement)",
            "config": {"workload": "configs[3] shape: %d synthetic %d-bp genomes per GPU sketched (-a %s -k %d), images all-gathered over "
                                   "RCCL (%d x %d B), every rank holds them as a resident set and computes its two bands of the lower triangle "
                                   "(%d of %d rows, equal printed pairs per rank)"
                                   % (G, L, algo, k, N, ib, sum(b - a for a, b in bands), N), "genomes_per_gpu": G, "genome_length": L, "algo": algo, "k": k, "p": p},
            "printed_pairs_per_s": pairs / elapsed, "stage_ms_rank0": stage, "ranks": evidence,
            "host_side": {"cardinalities_s_all_sketches": card_s, "rows_to_text_pairs_per_s_this_rank": host_pairs_per_s, "threads": threads,
                          "note": "lash_dist_rows + row text in C++ (liblash_host.so), outside the timed step: a full run's wall time is "
                                  "bounded by this, not by the GPU (profiles/r03/config3_full_one_gpu.log)"},
            "roofline": None, "cpu_baseline": None}))


def cli_bench(args):
    """--workload cli: the drop-in surface itself — `lash sketch -f list -o out` (lash_amd/bin/lash, the C++ host
    above the C ABI: main.rs:180-279, utils.rs:439-580) from FASTA FILES on tmpfs to `{o}_sketches.bin`, as a child process, timed on
    the wall clock by this process.  Never the default line: PCIe and the host's file reads are inside the step."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, ROOT)
    algo, k, p, L, G = args.algo, args.k, args.p if args.algo != "hmh" else 0, args.length, args.genomes
    assert L % 80 == 0, "--workload cli writes 80-column FASTA: --length must be a multiple of 80"
    exe = os.path.join(ROOT, "lash_amd", "bin", "lash")
    d = tempfile.mkdtemp(prefix="lash_bench_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        distinct = min(G, max(1, (6 << 30) // (L + L // 80)))          # at most ~6 GB of tmpfs; further paths are hard links (read in full all the same)
        # the files are written by a CHILD (the library's own generator on the GPU): this process never touches the GPU, it only
        # starts programs and reads the clock
        gen = ("import sys, os, numpy as np, torch\nsys.path.insert(0, %r)\nimport lash_amd\n"
               "ctx = lash_amd.Context(0)\nL, distinct, d = %d, %d, %r\n"
               "for g0 in range(0, distinct, 100):\n"
               "    n = min(100, distinct - g0)\n"
               "    dseq = torch.empty(n * L, dtype=torch.uint8, device='cuda')\n"
               "    ctx.synth_genomes_device(g0, n, L, dseq)\n"
               "    torch.cuda.synchronize()\n"
               "    host = dseq.cpu().numpy().reshape(n, L // 80, 80)\n"
               "    for i in range(n):\n"
               "        lines = np.empty((L // 80, 81), dtype=np.uint8)\n"
               "        lines[:, :80] = host[i]\n"
               "        lines[:, 80] = 10\n"
               "        with open(os.path.join(d, 'g%%d.fa' %% (g0 + i)), 'wb') as f:\n"
               "            f.write(b'>g%%d\\n' %% (g0 + i))\n"
               "            f.write(lines.tobytes())\n"
               "ctx.close()\n") % (ROOT, L, distinct, d)
        r = subprocess.run([sys.executable, "-c", gen], capture_output=True, text=True, timeout=3600)
        if r.returncode != 0:
            raise SystemExit("could not write the FASTA files: " + r.stderr[-2000:])
        names = []
        for g in range(G):
            if g >= distinct:
                os.link(os.path.join(d, "g%d.fa" % (g % distinct)), os.path.join(d, "g%d.fa" % g))
            names.append(os.path.join(d, "g%d.fa" % g))
        open(os.path.join(d, "list.txt"), "w").write("\n".join(names) + "\n")
        text_bytes = G * (L + L // 80 + len(">g0\n"))
        from lash_amd.shard import effective_cores            # (the cgroup's CPU quota, not the host's 256 logical CPUs: 64 reader threads on a
        threads = min(64, max(4, 2 * effective_cores()))      #  16-CPU quota cost a collection of small files a third of its rate)
        cmd = [exe, "sketch", "-f", os.path.join(d, "list.txt"), "-o", os.path.join(d, "out"), "-k", str(k), "-a", algo, "-t", str(threads)]
        if algo != "hmh":
            cmd += ["-p", str(p)]
        walls, inside = [], []
        for it in range(args.warmup + args.steps):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
            w = time.perf_counter() - t0
            if r.returncode != 0:
                raise SystemExit("lash sketch failed: " + r.stderr[-2000:])
            if it >= args.warmup:
                walls.append(w)
                for ln in (r.stdout + r.stderr).splitlines():
                    if ln.startswith("sketched ") and " in " in ln:
                        inside.append(float(ln.split(" in ")[1].split(" s")[0]))
        wall = sum(walls) / len(walls)
        kmers = G * (L - k + 1)
        print(json.dumps({
            "metric": "k-mers/s sketched (%s, k=%d) through the `lash sketch` command line" % (algo, k), "value": kmers / wall, "unit": "k-mers/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic FASTA files on tmpfs",
            "config": {"workload": "`lash sketch` END TO END: %d FASTA files of %d bp (80-column lines, %.2f GB of text; %d distinct, the rest hard links) on tmpfs "
                                   "-> {o}_sketches.bin + the two JSON files; process start, file reads, PCIe, device parse + pack + sketch, zstd and the write are "
                                   "all inside the step (a child process per step)" % (G, L, text_bytes / 1e9, distinct),
                       "genomes": G, "genome_length": L, "algo": algo, "k": k, "p": p, "threads": threads},
            "text_gb_per_s": text_bytes / wall / 1e9,
            "inside_sketch_files_s": (sum(inside) / len(inside)) if inside else None,
            "roofline": None, "cpu_baseline": None,
            "note": "not a kernel measurement: the step is bound by the host link (PCIe Gen5 x16 carries ~1 B per base) and the file reads; the resident-input "
                    "line is `python bench.py` without --workload"}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def visible_gpus():
    """GPUs this process' children could use, counted WITHOUT initialising HIP here (the parent of a self-launch must never touch the
    GPU): a one-shot CHILD asks torch (torch.cuda.device_count() does not initialise HIP on this image, and the child is gone before
    anything else starts).  The KFD topology is no substitute — a container that is handed one GPU of an 8-GPU node still lists eight
    nodes with SIMDs there (seen on the round-5 box).  None when the child cannot tell: let the ranks find out."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception:
        return None


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: start the N ranks ourselves — one process per
    GPU under torch.distributed.run, the command line the driver would have used — as a CHILD process, relay its output and
    return its exit code.  This process has not imported torch and never touches the GPU (no exec from a GPU-initialised
    process: the ranks are children of the launcher, which is a child of this).  The reference's counterpart is
    `files.par_iter()` (utils.rs:450-452): one task per shard, results in shard order."""
    import socket
    import subprocess
    # pre-flight: more ranks than GPUs would die inside every torchrun child with a traceback of its own; say it
    # once, in one line, before anything is started.  (LASH_BENCH_BACKEND=gloo is the launch dry run in which ranks SHARE devices.)
    have = visible_gpus()
    if have is not None and n > have and os.environ.get("LASH_BENCH_BACKEND", "nccl") == "nccl":
        print(json.dumps({"error": "bench.py --gpus %d: this process can see %d GPU%s (torch.cuda.device_count() in a child); nothing was launched"
                                   % (n, have, "" if have == 1 else "s"), "n_gpus_requested": n, "n_gpus_visible": have}))
        sys.stdout.flush()
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode          # stdout / stderr are inherited: rank 0's ONE JSON line passes through


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--genomes", type=int, default=12500, help="genomes per GPU (weak scaling).  12 500 x 5 Mbp = 62.5 GB of ASCII resident per GPU: "
                    "the north-star configuration (>= 10 000 genomes), and --gpus 8 is then exactly BASELINE configs[3]'s 100 000; "
                    "configs[1] is --genomes 1000")
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--algo", default="hmh")
    ap.add_argument("-k", type=int, default=16)
    ap.add_argument("-p", type=int, default=14)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--workload", choices=["genomes", "reads", "allpairs", "cli", "viral"], default="genomes",
                    help="genomes: --genomes x --length bp, one record each (configs[1]/[2]); reads: ONE sketch of --reads "
                         "150-bp records (configs[4] shape; use with --algo ull -p 12); allpairs: configs[3] shape — every rank "
                         "sketches --genomes genomes, the images are all-gathered (RCCL), every rank computes its block of "
                         "reference rows of the all-vs-all distance matrix; cli: `lash sketch` end to end from FASTA files on tmpfs; "
                         "viral: --genomes small genomes of unequal size (3..300 kbp, 1..4 records each), one sketch per genome — a virus / "
                         "plasmid / contig collection, whole through the persistent small-genome kernel (--length is ignored)")
    ap.add_argument("--reads", type=int, default=20_000_000, help="--workload reads: 150-bp records per step and GPU")
    ap.add_argument("--dirty", choices=["none", "nrun", "lower"], default="none",
                    help="nrun: one 100-byte run of N per genome; lower: every other 10 kb block lower-case (soft-masked "
                         "assembly: filter_out_n deletes those bytes, utils.rs:33-41)")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="CPU work of the cpu_baseline leg (the oracle on the host cores); the leg also "
                    "sketches 2 x cores genomes once to warm up and the same genomes as FASTA text for >= 1 s")
    ap.add_argument("--layout", default=None, help="lash_layout spec (include/lash_gfx950.h: 'hmh_x=low', 'kmer=lsb', 'codes=ACTG', 'hll_bucket=high', "
                    "comma-separated): the SAME workload under an alternative of the reference's unpinned crate rules (SURVEY App. D) — the kernels' "
                    "compile-time variants; the CPU leg and the parity spot check use the oracle with the same layout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-ubench", action="store_true", help="do not run tools/ubench_hash for the VALU ceiling (profiler runs: "
                    "rocprofv3 follows child processes); the committed profiles/valu.json figure is quoted instead")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.workload == "cli":                              # (cli: `lash sketch` from files on tmpfs, one GPU; see cli_bench)
        assert args.gpus == 1, "--workload cli runs the command line on one GPU (it takes -d for more)"
        if args.genomes == 12500:
            args.genomes = 1000                             # configs[1] size unless asked otherwise
        return cli_bench(args)

    import numpy as np
    import torch
    import lash_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # LASH_BENCH_BACKEND=gloo: a launch-logic dry run on fewer GPUs than ranks (ranks share devices, the barrier / max-time
        # reduction run on host tensors).  Never a scaling measurement; the JSON says so.
        backend = os.environ.get("LASH_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N > 1)"
    dev = torch.device("cuda", local_rank)

    algo, k, p, seed, L, G = args.algo, args.k, args.p if args.algo != "hmh" else 0, args.seed, args.length, args.genomes
    ib = lash_amd.image_bytes(algo, p)
    stream = torch.cuda.current_stream()
    ctx = lash_amd.Context(local_rank, stream=stream)     # raises without GPU / library: no fallback
    if args.layout:
        ctx.set_layout(args.layout)                        # every later call sketches under (and writes images in) this layout
        ib = ctx.image_bytes(algo, p)

    # ---- synthetic input, generated in HBM (the same generator as oracle/lash_oracle.c) ----
    reads = args.workload == "reads"
    if reads:
        # configs[4] shape: ONE sketch of N 150-bp records (a long synthetic sequence cut into reads)
        RL = 150
        n_rec, G, L = args.reads, 1, args.reads * RL
        assert L <= 0xFFFFFFFF - 64, "one call takes at most 2^32-64 bytes per sketch (stream larger inputs with F_ACCUMULATE)"
        first = 900_000 + rank
        d_seq = torch.empty(L, dtype=torch.uint8, device=dev)
        ctx.synth_genomes_device(first, 1, L, d_seq)
        rec_off = np.array([0, L], dtype=np.uint64)                      # byte offsets of the genome's first / last record
        goff = np.array([0, n_rec], dtype=np.uint64)
        d_rec = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * RL
        kmers_per_genome = n_rec * (RL - k + 1)
    elif args.workload == "viral":
        # round 5's shape (tools/viral_rate.py): a collection of SMALL genomes of unequal size — 3..300 kbp, log-uniform, 1..4 records each —,
        # one sketch per genome (utils.rs:450-509): whole through the persistent small-genome kernel.  --genomes of them per GPU
        # (200 000 = 12.9 GB is the size the round's numbers are quoted on); --length is ignored
        vrng = np.random.default_rng(13)                                # (the same lengths and cuts on every rank — the job's k-mer count is rank 0's times the ranks —, other bases)
        v_lens = np.exp(vrng.uniform(np.log(3e3), np.log(3e5), size=G)).astype(np.int64)
        v_gbo = np.concatenate([[0], np.cumsum(v_lens)]).astype(np.uint64)
        v_total = int(v_gbo[-1])
        v_nrec = vrng.integers(1, 5, size=G)
        cuts = (vrng.random(size=(G, 3)) * (v_lens[:, None] - 1)).astype(np.int64) + 1 + v_gbo[:-1, None].astype(np.int64)
        keep = np.arange(3)[None, :] < (v_nrec[:, None] - 1)
        v_rec = np.unique(np.concatenate([v_gbo[:-1].astype(np.int64), cuts[keep], [v_total]])).astype(np.uint64)
        first = 700_000 + rank
        d_seq = torch.empty(v_total, dtype=torch.uint8, device=dev)
        ctx.synth_genomes_device(first, 1, v_total, d_seq)               # one long synthetic sequence, cut into the genomes
        n_rec = len(v_rec) - 1
        rec_off = v_gbo                                                   # (the host array the entry takes: every genome's first byte)
        goff = np.searchsorted(v_rec, v_gbo).astype(np.uint64)
        d_rec = torch.from_numpy(v_rec.astype(np.int64)).to(dev)
        L = v_total // G                                                  # (the mean: labels only)
        kmers_per_genome = None
        viral_kmers = int(np.maximum(np.diff(v_rec.astype(np.int64)) - k + 1, 0).sum())
    else:
        n_rec = G
        d_seq = torch.empty(G * L, dtype=torch.uint8, device=dev)
        first = rank * G
        ctx.synth_genomes_device(first, G, L, d_seq)
        rec_off = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)        # one record per genome
        goff = np.arange(G + 1, dtype=np.uint64)
        d_rec = torch.from_numpy(rec_off.astype(np.int64)).to(dev)
        kmers_per_genome = L - k + 1
    torch.cuda.synchronize()
    surviving = L
    if args.dirty != "none":
        assert not reads, "--dirty applies to the genomes workload"
        ctx.synchronize()
        v = d_seq.view(G, L)
        if args.dirty == "nrun":
            # one run of 100 N per genome at a genome-dependent position (assembly gap)
            pos = (torch.arange(G, device=dev, dtype=torch.int64) * 2654435761 + 12345) % (L - 1000) + 500
            idx = pos[:, None] + torch.arange(100, device=dev, dtype=torch.int64)[None, :]
            v.scatter_(1, idx, torch.full(idx.shape, ord("N"), dtype=torch.uint8, device=dev))
            surviving = L - 100
        else:
            B = 10_000
            assert L % (2 * B) == 0
            v.view(G, L // B, B)[:, 1::2, :] |= 0x20                     # lower-case: deleted by filter_out_n
            surviving = L // 2
        kmers_per_genome = surviving - k + 1
        torch.cuda.synchronize()
    d_img = torch.zeros(G * ib, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    viral = args.workload == "viral"

    def alg_bytes_of(ascii_input):
        """SURVEY 8(d) for the whole batch of this rank: sum over genomes of (L or ceil(L / 4)) + S"""
        if viral:
            return int(v_total if ascii_input else ((v_lens + 3) // 4).sum()) + G * ib
        return G * algorithmic_bytes_per_genome(L, ib, ascii_input=ascii_input)

    if args.workload == "allpairs":
        allpairs_bench(args, ctx, torch, dist, dev, rank, world, algo, k, p, seed, L, G, d_seq, d_rec, goff, rec_off, d_img, ib)
        ctx.close()
        if dist:
            dist.destroy_process_group()
        return

    def step():
        ctx.sketch_batch_device(algo, k, p, seed, d_seq, d_rec, n_rec, goff, rec_off, d_img)

    # the GPU's clocks take some tens of milliseconds of load to settle after idle (the first launches run 6.4 ms, later
    # ones 5.2): a fixed untimed pre-heat, then the W warm-up steps the caller asked for, then exactly K timed steps
    for _ in range(12):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ctx.enable_timing(True)                   # HIP events on the stream the kernels run on
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tm = ctx.timing()
    ctx.enable_timing(False)
    evidence = rank_evidence(torch, dist, dev, rank, world, local_rank,
                             {"elapsed_ms": elapsed / args.steps * 1e3, "sketch_ms": tm["sketch_ms"] / max(tm["calls"], 1),
                              "kmers_census": int(tm["kmers"]), "first_genome": int(first)})
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # informational: the same batch kept resident as 2-bit (lash_pack_device once, then only the sketch + finalize stages)
    pk = ctx.pack_device(d_seq, d_rec, n_rec, goff, rec_off)
    ctx.sketch_packed_device(algo, k, p, seed, pk, d_img)
    torch.cuda.synchronize()
    ctx.enable_timing(True)
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.sketch_packed_device(algo, k, p, seed, pk, d_img)
    torch.cuda.synchronize()
    packed_elapsed = time.perf_counter() - tp0
    tm_pk = ctx.timing()                                    # (HIP events on the ctx stream: the packed-input sketch kernel's own time)
    ctx.enable_timing(False)
    pk.free()

    # oracle-free cross-check on every rank: the pack-first route (a different kernel chain) must give the same images
    # (timed with the same HIP events: gives the pack kernels' own HBM rate for the secondary roofline entry)
    d_img2 = torch.zeros_like(d_img)
    ctx.sketch_batch_device(algo, k, p, seed, d_seq, d_rec, n_rec, goff, rec_off, d_img2, flags=lash_amd.F_NO_DIRECT)
    torch.cuda.synchronize()
    ctx.enable_timing(True)
    for _ in range(3):
        ctx.sketch_batch_device(algo, k, p, seed, d_seq, d_rec, n_rec, goff, rec_off, d_img2, flags=lash_amd.F_NO_DIRECT)
    torch.cuda.synchronize()
    tm_pf = ctx.timing()
    ctx.enable_timing(False)
    routes_agree = bool(torch.equal(d_img, d_img2))
    del d_img2
    assert routes_agree, "direct and pack-first routes disagree"

    kmers_step_rank = viral_kmers if viral else G * kmers_per_genome
    assert tm["kmers"] == kmers_step_rank * args.steps, "device k-mer census disagrees with the workload"
    total_kmers = kmers_step_rank * world * args.steps
    value = total_kmers / elapsed

    out = None
    if rank == 0:
        # the dominant kernel: the direct (ASCII-reading) sketch kernel when the batch took that route (all of this
        # synthetic workload does), else the packed-input sketch kernel; both timed by HIP events on the ctx stream
        direct = tm["direct_launches"] > 0
        # (round 5) whole small genomes through the persistent kernel: ASCII in as well — L + S bytes per genome — and a name of its own
        sole = tm.get("sole_launches", 0) > 0 and not direct
        defer = tm["defer_launches"] > 0                 # HyperMinHash, long work items: signatures deferred
        stage_sketch_ms = tm["sketch_ms"] / max(tm["calls"], 1)
        # the dominant kernel's own time: the direct kernel's event bracket on clean input; on dirty input the genomes may be handed to
        # stream_sketch_kernel (and the optimistic pass skipped), so the figure is the whole sketch stage (direct + stream launches)
        dirty_in = args.dirty != "none"
        sketch_ms = tm["direct_ms"] / max(tm["calls"], 1) if direct and not dirty_in else stage_sketch_ms
        alg_bytes = alg_bytes_of(direct or sole)
        achieved = alg_bytes / (sketch_ms * 1e-3) / 1e9 if sketch_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-derived HBM bytes per sketch launch, if collected
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%s%s_k%d_p%d_g%d_l%d" % ("direct_" if direct else "", algo, k, p, G, L)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch") if args.dirty == "none" and not args.layout else None   # (the counter passes are of the clean workload under the default layout)
            except Exception:
                traffic = None
        out = {
            "metric": "k-mers/s sketched (%s, k=%d)" % (algo, k), "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic" if not dist or dist.get_backend() == "nccl" else "synthetic (LAUNCH DRY RUN: ranks share GPUs over gloo, not a scaling measurement)",
            "config": {"workload": ("ONE sketch of %d synthetic 150-bp records (%d bp) per GPU" % (n_rec, L) if reads else
                                    ("%d synthetic genomes of 3..300 kbp (log-uniform, mean %d bp, %.2f GB), 1..4 records each, per GPU%s" % (G, L, v_total / 1e9, "")) if viral else
                                    "%d synthetic %d-bp genomes per GPU%s" % (G, L, {"none": "", "nrun": ", one 100-byte N run in each",
                                                                                     "lower": ", every other 10 kb block lower-case"}[args.dirty]))
                                   + ", -a %s -k %d%s, seed %d, ASCII records resident in HBM -> sketch images in HBM "
                                     "(lash_sketch_batch_device: filter_out_n + k-mers + xxh3 + registers + images)"
                                   % (algo, k, "" if algo == "hmh" else " -p %d" % p, seed),
                       "genomes_per_gpu": G, "genome_length": L, "records_per_gpu": n_rec, "dirty": args.dirty,
                       "algo": algo, "k": k, "p": p, "sharding": "genomes across ranks", "layout": args.layout or "default"},
            "roofline": {"bound": "hbm", "kernel": ("sketch_kernel<DIRECT> + stream_sketch_kernel (sketch stage)" if dirty_in else ("sketch_kernel<DIRECT, DEFER>" if defer else "sketch_kernel<DIRECT>")) if direct or dirty_in else ("sole_sketch_kernel (whole genomes on persistent workgroups, ASCII in)" if sole else "sketch_kernel"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, committed (not collected in this run)" if traffic else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": sketch_ms,
                         "input": "ASCII records (1 B/base)" if direct or sole else "packed 2-bit words (0.25 B/base)",
                         # the OTHER accounting of SURVEY 8(d) / BASELINE.md (0.2566 B per k-mer at hmh k=16): the same genomes resident as the
                         # 2-bit stream, sketched by the packed-input kernel in this run — ceil(L/4) + S bytes per genome over that kernel's time
                         "frac_packed_accounting": (alg_bytes_of(False) / (tm_pk["sketch_ms"] / max(tm_pk["calls"], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                                   if tm_pk["sketch_ms"] > 0 else None,
                         "packed_accounting": {"algorithmic_bytes_per_launch": alg_bytes_of(False),
                                               "avg_launch_ms": tm_pk["sketch_ms"] / max(tm_pk["calls"], 1), "kernel": "sketch_kernel (packed 2-bit input, lash_sketch_packed_device)"},
                         "note": "integer-ALU/LDS-atomic bound kernel: see DESIGN.md 'Roofline'.  frac is on the bytes the dominant kernel really reads (ASCII, 1 B per base); "
                                 "frac_packed_accounting is the north-star's packed-2-bit accounting measured on the packed-input kernel"},
            "roofline_valu": valu_roofline(kmers_step_rank, sketch_ms, direct, algo, k, not args.no_ubench, defer and not dirty_in, p=p, reads=reads,
                                           sole=tm.get("sole_launches", 0) > 0),
            "stage_ms_per_step": {"pack": tm["pack_ms"] / max(tm["calls"], 1), "sketch": stage_sketch_ms,
                                  "finalize": tm["finalize_ms"] / max(tm["calls"], 1)},
            # what every rank measured and on which device (rank_evidence): per_rank_ms are the ranks' own wall clocks per step — `ms_per_step`
            # is their maximum —, ranks_seen the all_reduce(SUM) of ones through the backend, devices_distinct must equal n_gpus
            "ranks": evidence,
            "packed_resident_kmers_per_s_this_rank": kmers_step_rank * args.steps / packed_elapsed,   # 2-bit genomes kept in HBM
            "routes_agree": "direct and pack-first images identical (all %d genomes of this rank)" % G,
            # the pack-first route's own kernels (dirty genomes, raw FASTA/FASTQ input): the pack stage reads 1 B and writes
            # 0.25 B per base and is HBM-bound; its sketch kernel reads the packed form
            "pack_first_route": {
                "stage_ms_per_step": {"pack": tm_pf["pack_ms"] / 3, "sketch": tm_pf["sketch_ms"] / 3, "finalize": tm_pf["finalize_ms"] / 3},
                "pack_roofline": {"bound": "hbm", "kernel": "pack_lookback_kernel", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                  "achieved": G * (L + (L + 3) // 4) / (tm_pf["pack_ms"] / 3 * 1e-3) / 1e9,
                                  "frac": (alg_bytes_of(True) + alg_bytes_of(False) - 2 * G * ib) / (tm_pf["pack_ms"] / 3 * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }

    if rank == 0 and args.layout:
        out["roofline_valu"]["layout_note"] = ("--layout %s: the issue ceiling and the committed instruction count quoted here are the DEFAULT rule's "
                                               "(tools/ubench_hash, profiles/valu.json); profiles/r06/layout_risk.txt holds the measured ratio of every alternative" % args.layout)
    # ---- the CPU leg (rank 0, N = 1 only, outside the timed region): the oracle timed as the baseline and used as the
    #      checker of three of the images this run produced ----
    if rank == 0:
        if world == 1 and viral and not args.no_cpu_baseline:
            # parity: three genomes with their own records against the oracle; baseline: the oracle on equal-length stand-ins of the mean size
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            img = d_img.view(G, ib)
            ok = True
            if not args.no_parity_check:
                for g in sorted({0, G // 2, G - 1}):
                    host = d_seq[int(v_gbo[g]):int(v_gbo[g + 1])].cpu().numpy()
                    ro = (v_rec[int(goff[g]):int(goff[g + 1]) + 1] - v_gbo[g]).astype(np.uint64)
                    want = O.sketch_genomes(lash_amd.ALGOS[algo], k, p, seed, host, ro, np.array([0, len(ro) - 1], np.uint64),
                                            layout=O.parse_layout(args.layout) if args.layout else None)[0]
                    ok = ok and bool(np.array_equal(img[g].cpu().numpy(), want))
                out["parity_vs_oracle"] = "bit-identical (3 genomes with their records spot-checked)" if ok else "MISMATCH"
            out["cpu_baseline"], _ = cpu_baseline(algo, k, p, seed, int(L), args.cpu_seconds, first, {}, layout=args.layout)
            out["cpu_baseline"]["sample"] = "equal-length stand-ins of the collection's mean genome size (one record each): " + out["cpu_baseline"]["sample"]
            if not ok:
                print(json.dumps(out))
                raise SystemExit("parity check failed")
        elif world == 1 and not args.no_cpu_baseline and not reads and args.dirty == "none":
            img = d_img.view(G, ib)
            check = {g: img[g].cpu().numpy() for g in sorted({0, G // 2, G - 1})} if not args.no_parity_check else {}
            out["cpu_baseline"], ok = cpu_baseline(algo, k, p, seed, L, args.cpu_seconds, first, check, layout=args.layout)
            if check:
                out["parity_vs_oracle"] = "bit-identical (3 genomes spot-checked)" if ok else "MISMATCH"
            if not ok:
                print(json.dumps(out))
                raise SystemExit("parity check failed")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
