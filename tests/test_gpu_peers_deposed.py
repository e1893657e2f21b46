"""A deposed leader that LIVES and keeps pushing (missing #2 / N6): the receiver's fence on the replica-kernel path.
Rank 0's resident kernel is never parked; ranks 1 and 2 elect rank 1 among themselves, run a second stretch of rounds, and
only then does rank 0 push more rounds through the mappings it holds -- at the offsets the new term's entries occupy.  With
the fence (PeerMember.elect -> apus_gpu_fence_replica: a server that adopts a newer term leaves the ring and the mailbox the
old leader has mapped) the survivors' replicas are the oracle's bit for bit and the stale entries sit in the rings that were
left; without it (APUS_PEER_NO_RING_FENCE=1, diagnostic) the same walk damages both survivors' logs.
Reference: rc_revoke_log_access, src/dare/dare_ibv_rc.c:2156-2243.  Worker: tests/_peer_deposed_worker.py."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

from tests.test_gpu_peers import ROOT, _free_port

pytestmark = pytest.mark.gpu


def one_walk(ra, rb, rc, fence=True, n_send=1 << 16, timeout=300):
    out = os.path.join(tempfile.mkdtemp(), "res")
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APUS_DIST_BACKEND="gloo", APUS_DIST_ONE_DEVICE="1",
                   RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("APUS_PEER_NO_RING_FENCE", None)
        if not fence:
            env["APUS_PEER_NO_RING_FENCE"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_peer_deposed_worker.py"), out, str(n_send), str(ra), str(rb), str(rc)],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = {r: json.load(open(f"{out}.{r}")) for r in range(3) if os.path.exists(f"{out}.{r}")}
    return res, "\n".join(f"--- rank {r} stderr\n{o[1][-1500:]}" for r, o in enumerate(outs))


def test_a_deposed_leader_that_keeps_pushing_stores_into_rings_nobody_reads():
    for _ in range(2):
        res, err = one_walk(200, 150, 100)
        said = "\n".join(f"rank {r}: {v.get('error')}" for r, v in res.items() if not v["ok"])
        assert len(res) == 3 and all(v["ok"] for v in res.values()), f"ranks that reported: {sorted(res)}\n{said}\n{err}"
        assert res[0]["pushed"], "the deposed leader never pushed its stale rounds"
        for r in (1, 2):
            assert res[r]["fenced"] == 1 and res[r]["ring_bytes_differing"] == 0
            assert res[r]["retired_at_end_a"][1] == 2 and res[r]["retired_at_end_a"][2] == res[0]["stale_first_req_id"]


def test_without_the_fence_the_same_walk_damages_the_survivors_logs():
    """the diagnostic switch: what the fence is there for"""
    res, err = one_walk(200, 150, 100, fence=False)
    assert 0 in res and res[0]["pushed"], f"{res.get(0)}\n{err}"
    bad = [r for r in (1, 2) if r in res and not res[r]["ok"] and res[r].get("ring_bytes_differing", 0) > 0]
    assert bad, f"no survivor's log was damaged without the fence: {res}\n{err}"
