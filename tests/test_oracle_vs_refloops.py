"""CPU-only: the restated oracle (oracle/apus_oracle.c) against THE REFERENCE ITSELF.

oracle/_ref/libapus_ref_loops.so = /root/reference/src/dare/{dare_server,dare_ibv,dare_ibv_rc,
dare_ibv_ud,dare_ep_db,dare_kvs_sm}.c + config-dare.c + rbtree.c compiled UNMODIFIED (recipe:
oracle/Makefile `loops`) behind the in-process verbs / libev / libconfig stand-ins of
oracle/refshim/; one private copy per server, driven one polling() pass at a time by the trace.
Both sides replay the same events in lock step and are compared at every quiescent event:
all 8 log offsets of every server, every defined ring byte (incl. sender and reply[]), SID,
upcall counters and the upcall stream, prev_log_entry_head, and the leader's end/commit after
every pass.  This is the pin for the loop rows of SURVEY.md section 8 (a4, a6-a11, a14).

Skipped where neither the built library nor /root/reference exists; the same records are
committed as tests/golden/cluster_ref.json (tests/test_trace_oracle.py checks them everywhere)."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import refloops
from tests import traces
from tests.refparity import assert_same_state

pytestmark = pytest.mark.skipif(not refloops.available(), reason="oracle/_ref/libapus_ref_loops.so not available")


def lockstep(tr, check_at=("QUIESCE", "PRUNE")):
    oc = orc.Cluster(tr.group_size, tr.log_len)
    rc = refloops.RefCluster(tr.group_size, tr.log_len)
    reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
    try:
        for i, ev in enumerate(tr.events):
            op = ev[0]
            for c in (oc, rc):
                if op == "ROUND":
                    c.round(reqs[ev[1]:ev[1] + ev[2]], tr.arena)
                elif op == "ELECT":
                    c.elect(ev[1])
                elif op == "PRUNE":
                    c.tick_prune()
                elif op == "QUIESCE":
                    c.quiesce()
                elif op == "KILL":
                    c.kill(ev[1])
                elif op == "HOLD":
                    c.hold(ev[1])
                elif op == "RELEASE":
                    c.release(ev[1])
                elif op == "JOIN":
                    c.join(ev[1])
                else:
                    raise ValueError(ev)
            assert oc.n == rc.n
            if op in check_at:
                assert_same_state(oc, rc, oc.n, tag=f"{tr.name} event {i} {ev}")
                for r in range(oc.n):
                    if oc.alive(r) and not rc.gone(r):
                        cr = rc.cid(r); cr.pop("cid_offset")
                        assert oc.cid(r) == cr, f"{tr.name} event {i} {ev}: configuration of server {r}"
        assert_same_state(oc, rc, oc.n, tag=f"{tr.name} end")
        return oc, rc
    except BaseException:
        rc.close()
        raise


@pytest.mark.parametrize("name", sorted(traces.CATALOGUE))
def test_oracle_equals_reference(name):
    oc, rc = lockstep(traces.CATALOGUE[name]())
    rc.close()


def test_reference_boots_and_elects():
    """start-up as the reference does it: RC_SYN/SYNACK/ACK over UD multicast, every server a
    candidate of term 1, the winner's timeout fires first (term 2), blank CONFIG committed"""
    rc = refloops.RefCluster(5, 1 << 16)
    try:
        assert [rc.sid(i) for i in range(5)] == [(1 << 9) | i for i in range(5)]
        rc.elect(3)
        assert rc.leader == 3 and all(rc.sid(i) == (2 << 9) | (1 << 8) | 3 for i in range(5))
        for i in range(5):
            o = rc.log(i).offsets()
            assert (o["commit"], o["end"], o["apply"]) == (64, 64, 64)
        assert rc.cid(3)["bitmask"] == 0b11111
    finally:
        rc.close()


def test_fence_deposed_leader_cannot_write():
    """rc_revoke_log_access (dare_ibv_rc.c:2156): a server that voted in a new term keeps its
    LOG QP towards everybody but the new leader in RESET, so the deposed leader's log WRITEs to
    it are rejected and none of its entries land there."""
    tr = traces.steady3()
    rc = refloops.RefCluster(3, tr.log_len)
    reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
    try:
        rc.elect(0)
        rc.round(reqs[0:8], tr.arena)
        rc.quiesce()
        # servers 1 and 2 stop hearing the leader and elect 1 while 0 keeps running
        rc.hold(0)
        for i in (1, 2):
            while (rc.sid(i) & 0xFF) != i or (rc.sid(i) >> 8) & 1:
                rc.fire(i, 2)
        rc.fire(1, 2)
        rc.poll(2); rc.poll(1)
        assert (rc.sid(1) >> 8) & 1 and (rc.sid(1) & 0xFF) == 1          # 1 leads a higher term
        assert rc.sid(0) >> 9 < rc.sid(1) >> 9 and rc.leader == 0        # 0 has not noticed
        rc.hold(1)                                                        # only the path 0 -> 2 is open
        rc.release(0)
        before = rc.log(2).offsets()
        ring_before = rc.log(2).ring().copy()
        rc.round(reqs[8:16], tr.arena)          # the deposed leader appends and tries to replicate
        assert rc.log(0).offsets()["end"] == before["end"] + 8 * 128      # they are in ITS log ...
        assert rc.log(2).offsets() == before                              # ... and nowhere else
        assert np.array_equal(rc.log(2).ring(), ring_before)
        assert not rc.peer(2, 0)["log_access"] and rc.peer(2, 1)["log_access"]
    finally:
        rc.close()


def test_full_size_c2():
    """BASELINE configs[1] at full size: 2^20 x 64-byte entries through the 64 MiB ring (wraps,
    16 prune ticks), 3 servers"""
    from apus_amd import trace as T
    tr = T.config_c2()
    oc, rc = lockstep(tr, check_at=("QUIESCE",))
    assert oc.highest_rec(0) == (1 << 20) + 16
    rc.close()


def _random_join_trace(seed):
    """kills, joins into the freed slots and group extensions at random places of a stream of 128-byte
    entries (64-aligned: the reference's joiner survives its first persist pass, oracle/apus_oracle.c
    orc_join -8), prune ticks in between so that followers can dump their state machine again (-5)"""
    from apus_amd import trace as T
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3, 4, 5]))
    tr = T.steady_trace(n, 2400, 64, 4, 10, log_len=1 << 16, name=f"random_join_{seed}")
    ev, k, size, alive, dead = [], 0, n, set(range(n)), []
    next_evt = int(rng.integers(8, 30))
    for e in tr.events:
        ev.append(e)
        if e[0] != "ROUND":
            continue
        k += 1
        if k < next_evt:
            continue
        next_evt = k + int(rng.integers(25, 60))
        p = rng.random()
        if dead and p < 0.6:
            r = min(dead)                                  # the leader hands out the lowest empty slot
            ev += [("QUIESCE",), ("JOIN", r), ("QUIESCE",)]
            dead.remove(r); alive.add(r)
        elif p < 0.8 and size < 7 and not dead:
            ev += [("QUIESCE",), ("JOIN", size), ("QUIESCE",)]
            alive.add(size); size += 1
        elif len(alive) - 1 > size // 2 and len(alive) > 2:
            r = int(rng.choice(sorted(alive - {0})))
            ev += [("QUIESCE",), ("KILL", r), ("QUIESCE",)]
            alive.remove(r); dead.append(r)
    ev.append(("QUIESCE",))
    tr.events = ev
    return tr


@pytest.mark.parametrize("seed", range(16))
def test_random_joins_equal_reference(seed):
    tr = _random_join_trace(seed)
    try:
        orc.run_trace(tr)              # a dry run: a JOIN the reference cannot survive (-5 crash, -8 hang) is skipped
    except RuntimeError as e:
        pytest.skip(f"the oracle refuses this schedule ({e}): the reference itself crashes, hangs or leaves the "
                    "joiner retrying for ever there (orc_join's return codes)")
    oc, rc = lockstep(tr, check_at=("QUIESCE",))
    assert sum(1 for e in tr.events if e[0] == "JOIN") >= 1
    rc.close()


def _random_failure_trace(seed, wild=False):
    """followers cut off and released, followers killed, the leader killed and a reachable server elected,
    at random places of a stream with mixed entry sizes; a majority of the configured servers stays
    reachable.  Every release and every fail-over is followed by a quiescent event."""
    from apus_amd import trace as T
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([3, 5, 5, 7]))
    sizes = [(64,), (64, 107), (40, 64, 300), (100,)][int(rng.integers(0, 4))]
    tr = T.steady_trace(n, 1500, sizes, 6, [8, 16, (1, 24)][int(rng.integers(0, 3))], log_len=1 << 18,
                        name=f"random_failures_{seed}", seed=seed)
    ev, k = [], 0
    cfg, up, held, leader = set(range(n)), set(range(n)), set(), 0      # configured / alive / cut off
    next_evt = int(rng.integers(5, 20))
    for e in tr.events:
        ev.append(e)
        if e[0] != "ROUND":
            continue
        k += 1
        if k < next_evt:
            continue
        next_evt = k + int(rng.integers(6, 25))
        p = rng.random()
        followers = sorted((cfg & up) - {leader})
        reach = [f for f in followers if f not in held]
        need = n // 2 + 1                                   # cid.size[0] stays n: removal only clears bits
        if held and p < 0.35:
            f = int(rng.choice(sorted(held)))
            held.discard(f)
            ev += [("RELEASE", f), ("QUIESCE",)]
        elif p < 0.6 and (len(reach) + 1 > need or (wild and reach)):          # wild: the quorum may go for a while
            f = int(rng.choice(reach))
            held.add(f)
            ev += [("QUIESCE",), ("HOLD", f)]
        elif p < 0.75 and len(reach) + 1 > need and not held:
            f = int(rng.choice(reach))
            up.discard(f); cfg.discard(f)
            ev += [("QUIESCE",), ("KILL", f), ("QUIESCE",)]
        elif p < 0.9 and len(reach) > need and (wild or not held):
            # wild: servers that are cut off during the election fail both vote requests and are removed by the
            # new leader's first pass (check_failure_count, dare_server.c:1189-1230); they stay out
            w = int(rng.choice(reach))
            up.discard(leader); cfg.discard(leader)
            ev += [("QUIESCE",), ("KILL", leader), ("ELECT", w), ("QUIESCE",)]
            cfg -= held; up -= held; held.clear()
            leader = w
    for f in sorted(held):
        ev += [("RELEASE", f)]
    ev.append(("QUIESCE",))
    tr.events = ev
    return tr


@pytest.mark.parametrize("wild", [False, True])
@pytest.mark.parametrize("seed", range(12))
def test_random_failures_equal_reference(seed, wild):
    tr = _random_failure_trace(seed, wild)
    oc, rc = lockstep(tr, check_at=("QUIESCE",))
    rc.close()


def _random_mixed_trace(seed):
    """joins (freed slots, group extensions), kills, cut-offs and fail-overs -- incl. the election of a
    server that joined -- mixed at random; 128-byte entries (a joiner's first persist pass needs the alignment)"""
    from apus_amd import trace as T
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([3, 4, 5]))
    tr = T.steady_trace(n, 3000, 64, 4, [8, 10, 16][int(rng.integers(0, 3))], log_len=1 << 16,
                        name=f"random_mixed_{seed}", seed=seed)
    ev, k = [], 0
    size, cfg, held, dead, leader = n, set(range(n)), set(), [], 0
    next_evt = int(rng.integers(8, 25))
    for e in tr.events:
        ev.append(e)
        if e[0] != "ROUND":
            continue
        k += 1
        if k < next_evt:
            continue
        next_evt = k + int(rng.integers(20, 45))
        p = rng.random()
        followers = sorted(cfg - {leader})
        reach = [f for f in followers if f not in held]
        need = size // 2 + 1
        if held and p < 0.25:
            f = int(rng.choice(sorted(held)))
            held.discard(f)
            ev += [("RELEASE", f), ("QUIESCE",)]
        elif p < 0.40 and len(reach) + 1 > need:
            f = int(rng.choice(reach))
            held.add(f)
            ev += [("QUIESCE",), ("HOLD", f)]
        elif p < 0.60 and not held and (dead or size < 7):
            r = min(dead) if dead else size                     # the leader hands out the lowest empty slot
            ev += [("QUIESCE",), ("JOIN", r), ("QUIESCE",)]
            if dead:
                dead.remove(r)
            else:
                size += 1
            cfg.add(r)
        elif p < 0.78 and not held and len(reach) + 1 > need and len(cfg) > 2:
            f = int(rng.choice(reach))
            cfg.discard(f); dead.append(f)
            ev += [("QUIESCE",), ("KILL", f), ("QUIESCE",)]
        elif p < 0.92 and not held and len(reach) > need:
            w = int(rng.choice(reach))
            cfg.discard(leader); dead.append(leader)
            ev += [("QUIESCE",), ("KILL", leader), ("ELECT", w), ("QUIESCE",)]
            leader = w
    for f in sorted(held):
        ev += [("RELEASE", f)]
    ev.append(("QUIESCE",))
    tr.events = ev
    return tr


@pytest.mark.parametrize("seed", range(20))
def test_random_mixed_equal_reference(seed):
    tr = _random_mixed_trace(seed)
    try:
        orc.run_trace(tr)
    except RuntimeError as e:
        pytest.skip(f"the oracle refuses this schedule ({e}): the reference crashes, hangs or stalls there")
    oc, rc = lockstep(tr, check_at=("QUIESCE",))
    rc.close()
