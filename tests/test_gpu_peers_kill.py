"""The leader's PROCESS of a peer-mapped group dies with tickets in flight (missing #1 (b)): rank 0 issues a
long stretch of device-resident rounds through the replica kernels and `os._exit`s while most of its commands are still
queued -- no drain, no park.  The survivors park themselves, the one that holds more is elected on its device and leads a
second stretch; every survivor checks contiguity, that no commit doorbell promised more than a survivor holds in order,
and its own replica against an oracle that walks the schedule the crash really had.  Worker: tests/_peer_kill_worker.py
(plain processes, not torchrun: torchrun tears the group down as soon as one rank exits)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

from tests.test_gpu_peers import ROOT, _free_port

pytestmark = pytest.mark.gpu
WALKS = int(os.environ.get("APUS_PEER_KILLS", "3"))


def one_walk(n_send, ra, rb, grid=(1, 1), timeout=420):
    out = os.path.join(tempfile.mkdtemp(), "res")
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APUS_DIST_BACKEND="gloo", APUS_DIST_ONE_DEVICE="1",
                   RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_peer_kill_worker.py"), out, str(n_send), str(ra), str(rb),
                                       str(grid[0]), str(grid[1])], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = {r: json.load(open(f"{out}.{r}")) for r in (1, 2) if os.path.exists(f"{out}.{r}")}
    said = "\n".join(f"rank {r}: {v.get('error')}" for r, v in res.items() if not v["ok"])
    assert len(res) == 2 and all(v["ok"] for v in res.values()), (f"survivors: {sorted(res)}\n{said}\n" +
                                                                   "\n".join(f"--- rank {r} stderr\n{o[1][-2500:]}" for r, o in enumerate(outs)))
    assert procs[0].returncode == 0 and not os.path.exists(f"{out}.0")       # the leader left without a word
    return res[1]


def test_leader_process_dies_with_tickets_in_flight():
    """2^19 SEND entries staged, 6000 rounds issued as 64 commands, the leader gone after the first: WALKS walks, and at
    least one of them must have caught rounds in flight (the winner holds fewer rounds than were issued)."""
    seen = []
    for _ in range(WALKS):
        r = one_walk(1 << 19, 6000, 1000)
        seen.append((r["rounds_winner"], r["rounds_lag"], r["bells"], r["held"]))
    assert any(w < 6000 for w, _, _, _ in seen), f"no walk caught the leader with rounds in flight: {seen}"
    print(f"leader killed with tickets in flight, {WALKS} walks: (rounds the winner held, the other survivor held, commit doorbells, entry slots) = {seen}")
