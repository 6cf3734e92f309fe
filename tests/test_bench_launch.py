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
