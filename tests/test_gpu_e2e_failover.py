"""The leader's PROCESS is killed WITH ROUNDS IN FLIGHT (missing #1): redis-server processes under LD_PRELOAD, one
replica each (C host layer, APUS_GROUP_DIR), clients that keep SETting while `kill -9` hits the leader -- no drain, no idle
phase.  What the reference gets from in-order RC writes (dare_ibv_rc.c:1828-1863 walks old_end -> end, :1465-1643 posts the
log bytes before the end word) is what round 4's cumulative in-order ACKs and per-run doorbell numbering are for; here it is
TESTED:

 * no acknowledged request is lost: every SET whose reply a client received is in EVERY survivor's redis afterwards
   (a reply = committed by a majority + applied, proxy.c:160) -- after every kill and at the end;
 * the survivors' logs are identical and contiguous (idx without a gap, terms never fall);
 * they equal an oracle replay of the final leader's log under the schedule the elections really had: a server that held
   fewer entries than the winner when the old leader died is cut off (HOLD) behind the last entry it held, released for the
   election and caught up by the new leader (tests/_cluster.py:oracle_replay; the park points are the vote requests the
   servers left in the group directory).
(That a commit doorbell never promised more than a survivor holds in order is checked where the doorbell words can be
read: tests/test_gpu_peers_kill.py.)

APUS_KILL_CLUSTERS groups of APUS_KILL_N servers (default 10 x 5: two leaders killed per group = 20 kills)."""
import os
import time

import numpy as np
import pytest

from apus_amd import trace as T
from tests import _cluster as K
from tests.test_gpu_e2e_redis import REF, parse_dump

pytestmark = pytest.mark.gpu

CLUSTERS = int(os.environ.get("APUS_KILL_CLUSTERS", "10"))
N = int(os.environ.get("APUS_KILL_N", "5"))
LOG = 1 << 24


def one_group(n, stats):
    g = K.Group(n, log_len=LOG)
    acked_all = []
    elections = []
    try:
        g.start_all()
        leader, term = 0, 2
        kills = (n - 1) // 2
        for k in range(kills + 1):
            load = K.Load(g.ports[leader], 4, f"g{k}").start()
            t0 = time.time()
            want = 300 if k < kills else 400
            while load.n_acked() < want and time.time() - t0 < 30:
                time.sleep(0.005)
            assert load.n_acked() >= want, f"phase {k}: only {load.n_acked()} requests answered by server {leader}\n" + g.all_tails()
            if k == kills:
                acked_all += load.finish()
                break
            # ---- kill -9 while every connection has a request on the wire
            g.kill(leader)
            acked = load.finish()
            inflight = sum(1 for p in load.pending if p is not None)
            stats["inflight"] += inflight
            stats["kills"] += 1
            acked_all += acked
            term += 2
            c = g.wait_cfg(lambda c: c["term"] == term, 90)
            assert c is not None, f"no leader of term {term} after server {leader} was killed\n" + g.all_tails()
            parked = {i: v[2] for i, v in g.parked(term).items()}
            assert c["leader"] in parked and all(parked[i] <= parked[c["leader"]] for i in parked), (c, parked)
            if any(v < parked[c["leader"]] for v in parked.values()):
                stats["lagging"] += 1
            elections.append((term, c["leader"], leader, parked))
            leader = c["leader"]
            assert g.wait_log(leader, f"[T{term}] LEADER", 30)
            # ---- nothing a client was told is lost: on EVERY survivor
            for i in g.alive():
                missing = K.wait_keys(g.ports[i], acked_all, 30)
                assert not missing, (f"after kill {k} (term {term}, leader {leader}): {len(missing)} acknowledged SETs are missing in server {i}'s redis, "
                                     f"e.g. {missing[:3]}\n" + g.tail(i))
        survivors = sorted(g.alive())
        for i in survivors:
            missing = K.wait_keys(g.ports[i], acked_all, 30)
            assert not missing, f"at the end: {len(missing)} acknowledged SETs are missing in server {i}'s redis, e.g. {missing[:3]}"
        g.shutdown(leader)
    except BaseException:
        g.postmortem("failover_postmortem.txt")
        raise
    finally:
        g.close()
    assert os.path.exists(g.dumps[leader]), "no replica dump from the last leader\n" + g.tail(leader)
    reps, rings = parse_dump(g.dumps[leader], n)
    lead = reps[leader]
    assert lead["status"] == 0, reps
    assert lead["commit"] == lead["end"] == lead["apply"], lead
    ents = K.log_entries(rings[leader], lead["end"])
    # contiguous: idx without a gap from 1, terms never fall
    assert [e[0] for e in ents] == list(range(1, len(ents) + 1))
    assert all(a[1] <= b[1] for a, b in zip(ents, ents[1:]))
    assert {e[1] for e in ents} == set(range(2, term + 1, 2))
    for r in survivors:
        assert (reps[r]["commit"], reps[r]["end"]) == (lead["commit"], lead["end"]), (r, reps[r], lead)
    # every acknowledged SET is an entry of the log, in the order its connection sent it
    bodies = b"".join(e[5] for e in ents if e[2] == T.SEND)
    for key, val in acked_all[:: max(1, len(acked_all) // 200)]:
        assert f"{key} {val}".encode() in bodies, f"{key} was acknowledged but is not in the log"
    cl = K.oracle_replay(n, LOG, ents, elections)
    try:
        K.compare_with_oracle(cl, reps, rings, survivors)
    except AssertionError:
        g.postmortem("failover_postmortem.txt")
        raise
    stats["entries"] += len(ents)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "redis-server")), reason="oracle/_ref/redis-server not built (make -C oracle redis)")
def test_leader_killed_under_load_loses_nothing_acknowledged():
    stats = {"kills": 0, "inflight": 0, "lagging": 0, "entries": 0}
    for _ in range(CLUSTERS):
        one_group(N, stats)
    # the kills really met requests on the wire
    assert stats["inflight"] >= stats["kills"], stats
    print(f"{stats['kills']} leaders killed under load in {CLUSTERS} groups of {N}: {stats['inflight']} requests were on the wire, "
          f"{stats['lagging']} elections found a survivor behind the winner, {stats['entries']} log entries = oracle replay")
