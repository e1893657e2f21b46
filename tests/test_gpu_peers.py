"""One process per replica with peer-mapped logs (apus_amd/peers.py, the `bench.py --gpus N`
data plane) against the oracle: steady state, hold / release, no quorum, fail-over of the leader
PROCESS (BASELINE config 5: another rank takes over from what is in the control blocks), exact
fit.  Worker: tests/_peer_worker.py."""
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


def run_group(world, name, mode="per-call", flags=0, timeout=420, attempts=3):
    """Up to two retries when a rank vanished without a result (seen once in ~5 runs of 5 processes on a fresh box:
    a peer's gloo connection closes during start-up); what the lost rank wrote to stderr is kept under
    gpurun_out/ for the post-mortem.  A rank that REPORTS a mismatch fails the test at once."""
    for attempt in range(attempts):
        try:
            return _run_group(world, name, mode, flags, timeout)
        except _RankLost as e:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"peers_lost_{name}_{attempt}.err"), "w") as f:
                f.write(str(e))
            if attempt == attempts - 1:
                raise AssertionError(str(e))


class _RankLost(Exception):
    pass


def _run_group(world, name, mode, flags, timeout):
    out = os.path.join(tempfile.mkdtemp(), "res")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_peer_worker.py"), out, name, mode, str(flags)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APUS_DIST_BACKEND="gloo", APUS_DIST_ONE_DEVICE="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    res = [json.load(open(f"{out}.{r}")) for r in range(world) if os.path.exists(f"{out}.{r}")]
    if len(res) != world:
        said = "\n".join(f"rank {r['rank']}: {r.get('error')}" for r in res if not r["ok"])
        raise _RankLost(f"{world - len(res)} ranks produced no result; the others said:\n{said}\n{p.stdout[-2000:]}\n{p.stderr[-6000:]}")
    for r in res:
        if not r["ok"] and "Connection closed by peer" in str(r.get("error")) and all(
                q["ok"] or "Connection closed by peer" in str(q.get("error")) for q in res):
            raise _RankLost(f"rank {r['rank']}: {r.get('error')}\n{p.stderr[-6000:]}")
        assert r["ok"], f"rank {r['rank']}: {r.get('error')}"
    assert len({r["end"] for r in res if "end" in r}) >= 1
    return res


@pytest.mark.parametrize("name,world", [("steady3", 3), ("steady5_unaligned", 5), ("exact_fit", 3)])
def test_peer_mapped_group_matches_oracle(name, world):
    res = run_group(world, name)
    assert all(r["checks"] > 2 for r in res)
    assert [r["led"] for r in res] == [1] + [0] * (world - 1)


def test_peer_mapped_group_batched_launches():
    run_group(7, "steady7_mixed", mode="batched")


@pytest.mark.parametrize("name,world", [("hold_release", 5), ("no_quorum_wide", 3), ("kill_follower", 5)])
def test_peer_mapped_group_failures(name, world):
    run_group(world, name)


def test_peer_mapped_leader_process_fails_over():
    """config 5: the leader rank stops, rank 1 is elected and leads from its own process; later a
    follower is removed (benchmarks/reconf_bench.sh:249-343)"""
    res = run_group(5, "c5_failover")
    assert [r["led"] for r in res] == [1, 1, 0, 0, 0]


def test_peer_mapped_ack_aggregation_path():
    """per-entry ACK words + quorum scan (APUS_F_NO_FUSED_ACKS) across processes"""
    run_group(3, "steady3", flags=1)


def test_peer_mapped_two_failovers_truncate_a_divergent_log():
    """two leader PROCESSES die one after the other; the third leader's process casts the votes on its
    device, truncates server 1's divergent log through the mapping and replicates from there
    (tests/traces.py:double_failover_truncate, pinned on the reference)"""
    res = run_group(5, "double_failover_truncate")
    assert [r["led"] for r in res] == [1, 0, 1, 1, 0]


@pytest.mark.parametrize("name,world", [("join_upsize_3_to_5", 5), ("c5_rejoin", 5), ("join_then_failover", 4)])
def test_peer_mapped_group_join(name, world):
    """JOIN across processes (SURVEY.md 8 f2 + e): the joiner is a PROCESS of its own -- a spare rank, or the
    rank of a server that was killed and comes back as a new machine -- that zeroes the replica it hosts;
    the leader's engine recovers it on the device through the HIP IPC mappings (bulk transfer of log range +
    directory into the joiner's HBM: xGMI between GPUs), the group is extended 3 -> 4 -> 5 across processes,
    and in join_then_failover the joined process wins the next term and leads from its own engine.  Every
    rank checks its own replica from its own memory against the oracle."""
    res = run_group(world, name)
    assert all(r["checks"] >= 2 for r in res)


@pytest.mark.parametrize("name,world", [("steady3", 3), ("steady5_unaligned", 5), ("hold_release", 5), ("no_quorum_wide", 3),
                                        ("c5_failover", 5), ("join_then_failover", 4), ("steady7_mixed", 7)])
def test_peer_mapped_group_replica_kernels(name, world):
    """Every PROCESS runs the workgroups of the replica it hosts (apus_amd/csrc/apus_replica.h): the follower
    processes launch their follower kernels on their device, the leader's process its pipelined leader kernel; the
    leader pushes only log bytes + one doorbell per round through the HIP IPC mappings, every follower persists,
    writes its reply bytes into the leader's log and its round ACK into the leader's mailbox from ITS OWN kernel
    (R3), the leader commits by majority (popcount + ballot + count-trailing-ones over the round ACKs) and rings
    the commit doorbells (R4), every follower applies on its own.  Steady state, hold / release, no quorum, the
    fail-over of the leader process, JOIN and the election of the joined process: every rank checks its own
    replica from its own memory against the oracle at every quiescent event."""
    res = run_group(world, name, mode="replica")
    assert all(r["checks"] >= 2 for r in res)
