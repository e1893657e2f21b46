"""benchmarks/reconf_bench.sh:249-343 with real processes, in C: five redis-server processes under LD_PRELOAD, one replica
each (APUS_GROUP_DIR); bench -- kill the LEADER -- bench -- kill a FOLLOWER -- bench -- ADD a server -- bench.

 * the leader's process dies: the survivors elect (tests/test_gpu_e2e_cluster.py, test_gpu_e2e_failover.py);
 * a follower's process dies: the leader notices (its heartbeat look at the group directory), removes the server from the
   configuration with a CONFIG entry (check_failure_count, dare_server.c:1190-1230) and goes on with four... three servers;
 * a new machine starts with server_type=join (join_cluster_cb, dare_server.c:445-530): it is given the lowest empty place
   (the dead leader's, dare_ibv_ud.c:995-1021), the leader's device carries the JOIN out through the HIP-IPC mappings
   (apus_gpu_join: CONFIG entry, recovery of the joiner's log as one bulk transfer into ITS memory, its first persist /
   apply passes), the other members map the newcomer, and the newcomer's redis is brought up to date from the log it
   recovered.

Checked: after every phase every live redis holds every key that was acknowledged; at the end the logs of all live
servers -- the joined one included -- equal an oracle replay of ELECT 0, A, KILL 0, ELECT w, B, KILL f, C, JOIN 0, D."""
import os
import time

import numpy as np
import pytest

from apus_amd import trace as T
from tests import _cluster as K
from tests.test_gpu_e2e_redis import REF, parse_dump

pytestmark = pytest.mark.gpu
LOG = 1 << 24


def _phase(g, leader, tag, n_req, acked_all):
    load = K.Load(g.ports[leader], 4, tag).start()
    t0 = time.time()
    while load.n_acked() < n_req and time.time() - t0 < 60:
        time.sleep(0.005)
    got = load.finish()
    assert len(got) >= n_req, f"phase {tag}: only {len(got)} requests answered by server {leader}\n" + g.all_tails()
    acked_all += got
    for i in g.alive():
        missing = K.wait_keys(g.ports[i], acked_all, 30)
        assert not missing, f"phase {tag}: {len(missing)} acknowledged SETs are missing in server {i}'s redis, e.g. {missing[:3]}\n" + g.tail(i)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "redis-server")), reason="oracle/_ref/redis-server not built (make -C oracle redis)")
def test_kill_leader_kill_follower_add_server():
    from oracle import oracle as orc
    n = 5
    g = K.Group(n, log_len=LOG)
    acked = []
    try:
        g.start_all()
        _phase(g, 0, "a", 800, acked)
        # ---- the leader's process dies (idle: every survivor holds everything)
        g.kill(0)
        c = g.wait_cfg(lambda c: c["term"] == 4, 90)
        assert c is not None, "no leader of term 4\n" + g.all_tails()
        leader = c["leader"]
        assert g.wait_log(leader, "[T4] LEADER", 30)
        parked4 = {i: v[2] for i, v in g.parked(4).items()}
        _phase(g, leader, "b", 800, acked)
        # ---- a follower's process dies: removed by the leader
        victim = max(i for i in g.alive() if i != leader)
        seq0 = g.cfg_latest()["seq"]
        g.kill(victim)
        c = g.wait_cfg(lambda c: c["seq"] > seq0 and not (c["bitmask"] >> victim) & 1, 60)
        assert c is not None and c["kind"] == 2 and c["leader"] == leader, f"the leader did not remove server {victim}: {g.cfg_latest()}\n" + g.tail(leader)
        _phase(g, leader, "c", 800, acked)
        # ---- a new machine joins: the lowest empty place is the dead leader's
        seq0 = g.cfg_latest()["seq"]
        g.start(0, join=True)
        c = g.wait_cfg(lambda c: c["seq"] > seq0 and (c["bitmask"] >> 0) & 1, 120)
        assert c is not None and c["kind"] == 3, f"server 0 was not admitted: {g.cfg_latest()}\n" + g.all_tails()
        assert g.wait_log(0, "joined as server 0", 60), g.tail(0)
        from tests.test_gpu_e2e_redis import _wait_port
        assert _wait_port(g.ports[0], g.procs[0], timeout=60)
        _phase(g, leader, "d", 800, acked)                   # (the newcomer's redis holds phases a..c from its state transfer, d as a follower)
        live = sorted(g.alive())
        assert live == sorted({0, 1, 2, 3, 4} - {victim} - set()) and 0 in live
        g.shutdown(leader)
    except BaseException:
        g.postmortem("reconf_postmortem.txt")
        raise
    finally:
        g.close()
    assert os.path.exists(g.dumps[leader]), "no replica dump from the leader\n" + g.tail(leader)
    reps, rings = parse_dump(g.dumps[leader], n)
    lead = reps[leader]
    assert lead["status"] == 0 and lead["commit"] == lead["end"] == lead["apply"], reps
    ents = K.log_entries(rings[leader], lead["end"])
    assert [e[0] for e in ents] == list(range(1, len(ents) + 1))
    for r in live:
        assert (reps[r]["commit"], reps[r]["end"]) == (lead["commit"], lead["end"]), (r, reps[r], lead)

    # ---- the same history through the oracle: the CONFIG entries in the log say where the events were
    client = [e for e in ents if e[2] not in (T.CONFIG, T.HEAD, T.NOOP)]
    reqs = np.zeros(len(client), dtype=orc.REQ_DTYPE)
    arena = bytearray(16)
    for k, (idx, term, typ, rid, cid, body) in enumerate(client):
        reqs[k] = (rid, len(arena), cid, len(body), typ, (0, 0, 0))
        arena += body + bytes((-len(body)) % 16)
    arena = np.frombuffer(bytes(arena) + bytes(32), dtype=np.uint8)
    cfg_idx = [e[0] for e in ents if e[2] == T.CONFIG]
    # idx 1: blank CONFIG of term 2; two of term 4 (blank + removal of 0); one removal of the victim; one JOIN
    assert len(cfg_idx) == 5 and [e[1] for e in ents if e[2] == T.CONFIG] == [2, 4, 4, 4, 4], cfg_idx
    n_before = lambda idx: sum(1 for e in client if e[0] < idx)
    cuts = [n_before(cfg_idx[1]), n_before(cfg_idx[3]), n_before(cfg_idx[4])]
    cl = orc.Cluster(n, LOG)
    cl.elect(0)

    def feed(a, b):
        for g0 in range(a, b, 64):
            cl.round(reqs[g0:min(g0 + 64, b)], arena)
    feed(0, cuts[0]); cl.quiesce()
    cl.kill(0); cl.elect(leader)
    feed(cuts[0], cuts[1]); cl.quiesce()
    cl.kill(victim)
    feed(cuts[1], cuts[2]); cl.quiesce()
    cl.join(0); cl.quiesce()
    feed(cuts[2], len(reqs)); cl.quiesce()
    K.compare_with_oracle(cl, reps, rings, live)
    print(f"5 redis processes: leader killed (server {leader} elected, parked {parked4}), follower {victim} killed and removed, a new server joined "
          f"place 0; {len(acked)} acknowledged SETs in every live redis, {len(ents)} log entries = oracle replay")
