"""A leader that LIVES but does not answer: redis-server processes under LD_PRELOAD, one replica
each (C host layer, APUS_GROUP_DIR); the leader's process is SIGSTOPped under load -- its resident kernel keeps running on
the GPU, its mappings of the followers' log rings and mailboxes stay open.  The followers see no heartbeat, elect -- and with
EVERY election (round 6; not only behind a death) each of them LEAVES the ring and the mailbox the old leader has mapped
(apus_gpu_fence_replica: the receiver's side of rc_revoke_log_access, dare_ibv_rc.c:2156-2243) and maps the others' new ones
(apus_gpu_remap_fenced) before anything is voted on.  Then the old leader is SIGCONTed: it finds a newer term's announcement
and steps down; whatever it pushed in between went into memory nobody reads.

Checked: every SET a client was answered -- by the old leader before the stop, by the new one after -- is in both
survivors' redis; the survivors' logs are identical, contiguous, and equal an oracle replay of the schedule (the old leader
removed by the new term's first pass, like a dead one: check_failure_count, dare_server.c:1189-1230); both survivors fenced;
the old leader said it was deposed."""
import os
import signal
import time

import pytest

from apus_amd import trace as T
from tests import _cluster as K
from tests.test_gpu_e2e_redis import REF, parse_dump

pytestmark = pytest.mark.gpu

LOG = 1 << 24


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "redis-server")), reason="oracle/_ref/redis-server not built (make -C oracle redis)")
def test_a_stopped_leader_is_deposed_and_the_followers_leave_what_it_has_mapped(monkeypatch):
    monkeypatch.setenv("APUS_HB_TIMEOUT_MS", "700")
    n = 3
    g = K.Group(n, log_len=LOG)
    acked_all = []
    try:
        g.start_all()
        load = K.Load(g.ports[0], 4, "p0").start()
        t0 = time.time()
        while load.n_acked() < 300 and time.time() - t0 < 30:
            time.sleep(0.005)
        assert load.n_acked() >= 300, g.all_tails()
        # ---- the leader's process stops answering, with requests on the wire; its kernel stays resident
        g.procs[0].send_signal(signal.SIGSTOP)
        time.sleep(0.2)
        acked_all += load.finish(timeout=5)
        c = g.wait_cfg(lambda c: c["term"] == 4, 90)
        assert c is not None and c["leader"] in (1, 2), f"no leader of term 4 while server 0 was stopped: {c}\n" + g.all_tails()
        leader = c["leader"]
        assert not (c["bitmask"] & 1), c                                  # (server 0 is out of the configuration: it did not answer the votes)
        assert g.wait_log(leader, "[T4] LEADER", 30), g.all_tails()
        parked = {i: v[2] for i, v in g.parked(4).items()}
        assert set(parked) == {1, 2}, parked
        for i in (1, 2):
            assert g.wait_log(i, "no heartbeat", 5), g.tail(i)            # the election was NOT behind a death ...
            assert g.wait_log(i, "left the log ring and the mailbox server 0 has mapped (fence 1)", 5), g.tail(i)     # ... and fenced all the same
        # ---- the new term carries traffic; nothing a client was told is lost
        load2 = K.Load(g.ports[leader], 4, "p1").start()
        t0 = time.time()
        while load2.n_acked() < 300 and time.time() - t0 < 30:
            time.sleep(0.005)
        assert load2.n_acked() >= 300, g.all_tails()
        # ---- the old leader comes back: it must step down, not lead
        g.procs[0].send_signal(signal.SIGCONT)
        assert g.wait_log(0, "deposed: server %d leads term 4" % leader, 30), g.tail(0)
        time.sleep(0.3)
        acked_all += load2.finish()
        for i in (1, 2):
            missing = K.wait_keys(g.ports[i], acked_all, 30)
            assert not missing, f"{len(missing)} acknowledged SETs are missing in server {i}'s redis, e.g. {missing[:3]}\n" + g.tail(i)
        g.procs[0].kill()
        g.procs[0].wait(timeout=30)
        g.shutdown(leader)
    except BaseException:
        g.postmortem("partition_postmortem.txt")
        raise
    finally:
        g.close()
    reps, rings = parse_dump(g.dumps[leader], n)
    lead = reps[leader]
    assert lead["status"] == 0, reps
    assert lead["commit"] == lead["end"] == lead["apply"], lead
    ents = K.log_entries(rings[leader], lead["end"])
    assert [e[0] for e in ents] == list(range(1, len(ents) + 1))          # contiguous
    assert {e[1] for e in ents} == {2, 4}
    other = 3 - leader
    assert (reps[other]["commit"], reps[other]["end"]) == (lead["commit"], lead["end"]), (reps[other], lead)
    bodies = b"".join(e[5] for e in ents if e[2] == T.SEND)
    for key, val in acked_all[:: max(1, len(acked_all) // 200)]:
        assert f"{key} {val}".encode() in bodies, f"{key} was acknowledged but is not in the log"
    cl = K.oracle_replay(n, LOG, ents, [(4, leader, 0, parked)])
    K.compare_with_oracle(cl, reps, rings, [1, 2])
