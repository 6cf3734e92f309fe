"""All-vs-all `dist` over several ranks with REAL GPU work (BASELINE configs[3] in miniature; lash_amd/allpairs.py): two
worker processes share GPU 0 — each sketches its shard of the files with the HIP library, the images are all-gathered (gloo,
because two ranks cannot form an RCCL communicator on one device), each rank computes its block of reference rows with the
pair kernels — and the concatenated output must be byte-identical to the single-process C++ `lash sketch` + `lash dist`.
On an 8-GPU node the same module runs with --backend nccl, one GPU per rank (the driver's SCALE run covers `bench.py`)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import host_lib as H
import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("algo,p,k,extra", [("hmh", 10, 16, []), ("hll", 10, 21, ["--dm"]), ("ull", 12, 16, ["-e", "ml", "-m", "0"])])
def test_two_ranks_on_one_gpu_equal_the_cli(tmp_path, algo, p, k, extra):
    rng = np.random.default_rng(5)
    base = O.synth_genome(700, 200_000)
    paths = []
    for i in range(7):
        g = base.copy() if i < 5 else O.synth_genome(700 + i, 150_000 + 1000 * i)
        idx = rng.random(len(g)) < (0.0, 0.002, 0.01, 0.03, 0.1, 0, 0)[i]
        g[idx] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(idx.sum()))
        f = tmp_path / ("g%d.fa" % i)
        f.write_bytes(b">g%d\n" % i + b"\n".join(g.tobytes()[j:j + 80] for j in range(0, len(g), 80)) + b"\n")
        paths.append(str(f))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["-f", "list.txt", "-a", algo, "-k", str(k), "-p", str(p)]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "-m", "lash_amd.allpairs", "--backend", "gloo", "--device", "0", "-o", "multi.tsv"]
                       + args + extra, cwd=tmp_path, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Distances computed." in r.stdout
    r = subprocess.run([H.CLI, "sketch", "-o", "one"] + args, cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([H.CLI, "dist", "-q", "one", "-r", "one", "-o", "one.tsv", "--file-order"] + extra, cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    multi, one = (tmp_path / "multi.tsv").read_text(), (tmp_path / "one.tsv").read_text()
    assert multi == one                                          # same rows, same order (file order), same digits
    assert len(one.strip().split("\n")) == (8 if "--dm" in extra else 1 + 7 * 8 // 2)
