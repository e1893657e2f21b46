"""Multi-process replica group (one process per replica, the --gpus N path of
bench.py) against the oracle.  The GPU box has a single device, so every rank uses
GPU 0 and the exchange is staged through gloo; on a multi-GPU node the same code
runs with backend nccl (RCCL) and device tensors."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_send,log_len", [(2, 1500, 1 << 17), (3, 2500, 1 << 18)])
def test_group_of_processes_matches_oracle(world, n_send, log_len):
    out = os.path.join(tempfile.mkdtemp(), "res")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_group_worker.py"), out, str(n_send), str(log_len)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    res = []
    for r in range(world):
        path = f"{out}.{r}"
        assert os.path.exists(path), f"rank {r} produced no result\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}"
        res.append(json.load(open(path)))
    for r in res:
        assert r["ok"], f"rank {r['rank']}: {r.get('error')}\n{p.stderr[-2000:]}"
    assert len({r["end"] for r in res}) == 1
