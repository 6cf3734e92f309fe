"""bench.py's multi-rank launch logic on real hardware ("--gpus N has never executed with N > 1 anywhere"):
the driver's exact command line with N = 2, as a dry run on ONE GPU — LASH_BENCH_BACKEND=gloo lets the two ranks share
device 0, the barrier and the max-over-ranks time reduction run on host tensors.  Checks rank / genome-id arithmetic, the
census of both ranks, the JSON contract.  It is not a scaling measurement and the JSON line says so."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["unit"] == "k-mers/s"
    assert j["cpu_baseline"] is None                            # N > 1: no CPU leg
    assert "DRY RUN" in j["data"]
    kmers = 2 * 40 * (5_000_000 - 16 + 1) * 3                    # both ranks' genomes, every step
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 * 3 / kmers - 1) < 1e-6
    assert j["roofline"]["frac"] > 0 and j["roofline_valu"]["bound"] == "valu-issue"
    _check_ranks(j, 2)


def _check_ranks(j, n):
    """the line carries what every rank measured and on which device, and a count that went THROUGH the backend."""
    r = j["ranks"]
    assert r["ranks_seen"] == n == r["ranks_expected"] and len(r["per_rank_ms"]) == n and len(r["devices"]) == n
    assert [q["rank"] for q in r["per_rank"]] == list(range(n)) and len({q["pid"] for q in r["per_rank"]}) == n
    assert all(q["hostname"] and q["device"]["id"] for q in r["per_rank"])
    assert r["imbalance"] >= 1.0 and abs(max(r["per_rank_ms"]) / j["ms_per_step"] - 1) < 0.05       # ms_per_step = the slowest rank's
    assert 1 <= r["devices_distinct"] <= n                      # (the dry run's ranks share a GPU; on a real node it must equal n: docs/SCALE_RUNBOOK.md)


def test_two_ranks_dry_run_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, LASH_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--genomes", "40"],
                       cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    _check_line(r)


def test_plain_command_line_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun around it, the way the driver runs --gpus 1) starts its ranks
    itself as a child process and prints exactly one JSON line with n_gpus = 2."""
    env = dict(os.environ, LASH_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--genomes", "40"],
                       cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    _check_line(r)


def _one_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_every_workload_rehearses_with_two_ranks():
    """the first real multi-GPU run must not be lost to a trivial failure, so every workload's multi-rank code path
    (sharding, barrier, max-over-ranks, and for allpairs the gather of the images and the row bands) runs here with two ranks on one GPU
    over gloo: exactly one JSON line, n_gpus = 2, marked as a dry run."""
    env = dict(os.environ, LASH_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "reads", "--algo", "ull", "-p", "12",
                        "--reads", "2000000"], cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    j = _one_line(r)
    assert j["n_gpus"] == 2 and "DRY RUN" in j["data"] and j["cpu_baseline"] is None
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 * 2 / (2 * 2_000_000 * (150 - 16 + 1) * 2) - 1) < 1e-6
    _check_ranks(j, 2)
    assert [q["kmers_census"] for q in j["ranks"]["per_rank"]] == [2_000_000 * (150 - 16 + 1) * 2] * 2      # every rank's device census
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "allpairs", "--genomes", "300",
                        "--length", "200000"], cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    j = _one_line(r)
    assert j["n_gpus"] == 2 and "DRY RUN" in j["data"] and j["printed_pairs_per_s"] > 0
    assert set(j["stage_ms_rank0"]) == {"sketch", "gather", "set + pairs"}
    _check_ranks(j, 2)
    g = j["ranks"]["gather"]
    assert g["bytes_received_per_rank"] == 300 * 32768 and g["xgmi_peak_GBps_per_gpu"] == 7 * 153.0
    assert all(q["rows"] > 0 and q["gather_stage_ms"] >= 0 for q in j["ranks"]["per_rank"])
    assert sum(q["rows"] for q in j["ranks"]["per_rank"]) == 600                                       # the bands cover the matrix


def test_asking_for_more_gpus_than_the_box_has_fails_in_one_line():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LASH_BENCH_BACKEND"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1"], cwd=ROOT, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 2, r.stdout[-1000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus_requested"] == 8 and "Traceback" not in r.stderr


def test_the_cli_workload_prints_the_same_contract():
    """`bench.py --workload cli` times `lash sketch` itself — FASTA files on tmpfs in, sketches.bin out, a child
    process per step — and prints the usual JSON line (never the default line)."""
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LASH_BENCH_BACKEND"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "cli", "--genomes", "60", "--length", "400000", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    j = _one_line(r)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "k-mers/s" and "END TO END" in j["config"]["workload"]
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 / (60 * (400000 - 16 + 1)) - 1) < 1e-6 and j["text_gb_per_s"] > 0


def test_the_viral_workload_prints_the_same_contract():
    """`bench.py --workload viral`: round 5's shape — small genomes of unequal size, 1..4 records each, whole through the persistent kernel —
    with the usual line: byte accounting over the real lengths, the kernel named, three genomes with their records against the oracle."""
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LASH_BENCH_BACKEND", "LASH_SOLE_MAX"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "viral", "--genomes", "3000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"],
                       cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    j = _one_line(r)
    assert j["n_gpus"] == 1 and j["unit"] == "k-mers/s" and "3..300 kbp" in j["config"]["workload"]
    assert j["roofline"]["kernel"].startswith("sole_sketch_kernel") and j["roofline"]["input"].startswith("ASCII")
    assert j["parity_vs_oracle"].startswith("bit-identical") and j["cpu_baseline"]["value"] > 0
    assert j["roofline"]["algorithmic_bytes_per_launch"] > 3000 * 32768 and 0 < j["roofline"]["frac"] < 1


def test_the_layout_flag_reproduces_the_line_on_an_alternative_rule():
    """`bench.py --layout SPEC` runs the same workload under an alternative of the reference's unpinned crate rules — the kernels'
    compile-time variants — and its parity spot check and CPU leg use the oracle with the same layout."""
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LASH_BENCH_BACKEND"):
        env.pop(v, None)
    for spec in ("hmh_x=low", "kmer=lsb"):
        r = subprocess.run([sys.executable, "bench.py", "--genomes", "300", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--layout", spec],
                           cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
        j = _one_line(r)
        assert j["config"]["layout"] == spec and j["parity_vs_oracle"].startswith("bit-identical"), j.get("parity_vs_oracle")
        assert j["roofline"]["kernel"] == "sketch_kernel<DIRECT, DEFER>" and j["roofline"]["traffic"] is None      # x = low defers like the default
        assert j["ranks"]["ranks_seen"] == 1 and j["ranks"]["devices_distinct"] == 1 and "layout_note" in j["roofline_valu"]
        assert abs(j["value"] * j["ms_per_step"] * 1e-3 / (300 * (5_000_000 - 16 + 1)) - 1) < 1e-6
