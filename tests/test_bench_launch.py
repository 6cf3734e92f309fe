"""bench.py --gpus N without a rendezvous in the environment starts its own ranks.  CPU side: the command it
builds is the driver's torch.distributed.run line, it is started as a child (never exec), the parent imports neither torch nor
the library, and the child's exit code is the parent's."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_builds_the_drivers_command(monkeypatch):
    b = _bench()
    seen = {}

    def fake_run(cmd, env=None, cwd=None, **kw):
        seen["cmd"], seen["env"], seen["cwd"] = cmd, env, cwd
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        b.main()
        raise AssertionError("main() must exit with the child's code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["cwd"] == ROOT


def test_parent_of_a_self_launch_never_loads_torch():
    """Run the real thing with a stand-in for `python -m torch.distributed.run` on PYTHONPATH: the parent must not have imported
    torch (a GPU-initialised parent must not spawn ranks on this pool), and the ranks' output passes through."""
    import tempfile
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "torch", "distributed"))
    open(os.path.join(d, "torch", "__init__.py"), "w").write("")
    open(os.path.join(d, "torch", "distributed", "__init__.py"), "w").write("")
    open(os.path.join(d, "torch", "distributed", "run.py"), "w").write(
        "import sys, json\nprint(json.dumps({'argv': sys.argv[1:]}))\nsys.exit(3)\n")
    env = dict(os.environ, PYTHONPATH=d)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "allpairs", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 3, r.stderr[-2000:]
    import json
    argv = json.loads(r.stdout.strip().splitlines()[-1])["argv"]
    assert argv[-6:] == ["--gpus", "2", "--workload", "allpairs", "--steps", "1"] and "--nproc-per-node" in argv


def test_more_ranks_than_gpus_is_refused_in_one_line(monkeypatch, capsys):
    """`bench.py --gpus 8` on a node with fewer GPUs used to die inside every torchrun child with a traceback per
    rank.  The parent now counts the GPUs without touching HIP (KFD topology, or a one-shot child) and says so once: one JSON line, exit
    code 2, nothing launched."""
    import json
    b = _bench()
    launched = []
    monkeypatch.setattr(b, "visible_gpus", lambda: 1)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: launched.append(a) or subprocess.CompletedProcess(a, 0))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LASH_BENCH_BACKEND", raising=False)
    try:
        b.main()
        raise AssertionError("main() must exit")
    except SystemExit as e:
        assert e.code == 2
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(out) == 1 and not launched
    j = json.loads(out[0])
    assert j["n_gpus_requested"] == 8 and j["n_gpus_visible"] == 1 and "nothing was launched" in j["error"]
    # the dry run (ranks share devices over gloo) is not refused
    monkeypatch.setenv("LASH_BENCH_BACKEND", "gloo")
    try:
        b.main()
    except SystemExit as e:
        assert e.code == 0 and launched


def _evidence_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = _bench()
        ev = b.rank_evidence(torch, dist, torch.device("cpu"), rank, world, rank, {"elapsed_ms": 10.0 + rank, "sketch_ms": 9.0, "kmers_census": 7 * (rank + 1)})
        q.put((rank, ev))
    finally:
        dist.destroy_process_group()


def test_rank_evidence_over_gloo_world_size_3():
    """every rank's own time, host, process and device identity reach the JSON line through the backend, with a count
    (all_reduce SUM of ones) that proves how many ranks the communicator joined.  World size 3 over gloo here; nccl (= RCCL) on the node."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_evidence_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in range(3):
        ev = got[r]
        assert ev["backend"] == "gloo" and ev["ranks_seen"] == 3 == ev["ranks_expected"]
        assert ev["per_rank_ms"] == [10.0, 11.0, 12.0] and abs(ev["imbalance"] - 1.2) < 1e-12
        assert [x["rank"] for x in ev["per_rank"]] == [0, 1, 2] and [x["kmers_census"] for x in ev["per_rank"]] == [7, 14, 21]
        assert len({x["pid"] for x in ev["per_rank"]}) == 3 and len(ev["devices"]) == 3
    # one rank, no process group: the same fields
    import torch
    one = _bench().rank_evidence(torch, None, torch.device("cpu"), 0, 1, 0, {"elapsed_ms": 5.0})
    assert one["ranks_seen"] == 1 and one["per_rank_ms"] == [5.0] and one["imbalance"] == 1.0 and one["devices_distinct"] == 1
