"""Named event traces shared by the parity tests: the same catalogue is replayed on
(a) the reference itself (oracle/_ref/libapus_ref_loops.so, tests/test_oracle_vs_refloops.py
and tests/golden/make_cluster_golden.py), (b) the restated oracle (tests/test_trace_oracle.py
against the committed golden records) and (c) the GPU engine (tests/test_gpu_parity.py)."""
from __future__ import annotations

from apus_amd import trace as T


def _with_events(base, inserts, drop_prune=False):
    """inserts: {k: [events]} placed behind the k-th ROUND event (1-based)."""
    ev, k = [], 0
    for e in base.events:
        if drop_prune and e[0] == "PRUNE":
            continue
        ev.append(e)
        if e[0] == "ROUND":
            k += 1
            ev += inserts.get(k, [])
    base.events = ev
    return base


def steady3():
    return T.steady_trace(3, 2000, 64, 8, 64, log_len=1 << 16, name="steady3")


def steady5_unaligned():
    return T.steady_trace(5, 3000, 100, 8, 32, log_len=1 << 16, name="steady5_unaligned")


def steady7_mixed():
    return T.steady_trace(7, 3000, (64, 128, 256, 512, 1024), 16, (1, 64), log_len=1 << 18, name="steady7_mixed")


def c2_small():
    return T.config_c2(n_send=1 << 14, log_len=1 << 20)


def c3_small():
    return T.config_c3(n_send=1 << 12, log_len=1 << 21)


def c4_small():
    return T.config_c4(n_send=1 << 12, log_len=1 << 21)


def c5_failover():
    return T.config_c5(per_phase=600, log_len=1 << 18, batch=16)


def hold_one_of_three():
    tr = T.steady_trace(3, 200, 64, 4, 10, log_len=1 << 16, name="hold_one_of_three")
    return _with_events(tr, {2: [("HOLD", 2)]}, drop_prune=True)


def hold_release():
    tr = T.steady_trace(5, 2000, 64, 8, 32, log_len=1 << 19, name="hold_release")
    return _with_events(tr, {10: [("HOLD", 2)], 20: [("RELEASE", 2), ("QUIESCE",)],
                             30: [("HOLD", 4), ("HOLD", 1)],
                             36: [("RELEASE", 4), ("RELEASE", 1), ("QUIESCE",)]}, drop_prune=True)


def no_quorum():
    tr = T.steady_trace(3, 400, 64, 4, 16, log_len=1 << 16, name="no_quorum")
    return _with_events(tr, {5: [("HOLD", 1), ("HOLD", 2)], 12: [("RELEASE", 1), ("QUIESCE",)]}, drop_prune=True)


def no_quorum_prune():
    tr = T.steady_trace(3, 1200, 64, 4, 16, log_len=1 << 18, prune_bytes=8 << 10, name="no_quorum_prune")
    return _with_events(tr, {6: [("HOLD", 1), ("HOLD", 2)], 40: [("RELEASE", 1), ("RELEASE", 2), ("QUIESCE",)]})


def exact_fit():
    return T.steady_trace(3, 1024, 64, 1, 8, log_len=1 << 14, prune_bytes=(1 << 14) // 4, name="exact_fit")


def kill_follower():
    tr = T.steady_trace(5, 400, 64, 4, 10, log_len=1 << 16, name="kill_follower")
    return _with_events(tr, {9: [("KILL", 3)]})


def evict_slow_follower():
    """one of three followers is cut off and stays so: the prune ticks cannot move the head past its apply
    offset, the log fills to 75 % and force_log_pruning (dare_server.c:2069-2122) removes it from the
    configuration; the other two go on and the ring wraps over where it was stuck"""
    tr = T.steady_trace(3, 1200, 64, 4, 10, log_len=1 << 16, name="evict_slow_follower")
    return _with_events(tr, {6: [("QUIESCE",), ("HOLD", 2)], 60: [("QUIESCE",)]})


def wrap_quirk_second_round():
    """a round boundary lands on len - 64: the next round's first entry leaves a stale header there and goes
    to offset 0 (case-2 wrap) while the commit pointer is parked on that position -- the pass "commits"
    offset 0 (dare_ibv_rc.c:1725-1758) and, because it returned before the end doorbell went out, the pass
    BEHIND it commits only the wrapped round and returns again: end / commit after the three passes are
    (1280, 0), (2560, 1280), (3840, 3840)"""
    return T.steady_trace(3, 1200, 64, 16, 10, log_len=1 << 16, name="wrap_quirk_second_round")


def park_commit_at_wrap():
    """Case-1 wrap (the header does not fit: 28 bytes left) with the commit pointer parked on it
    and NO quorum: update_remote_logs "commits" offset 0 (dare_ibv_rc.c:1725-1758) and
    apply_committed_entries then applies the first entry of the new lap although it is not
    committed (log_get_entry redirects log->apply in place, dare_log.h:327-330) -- the client is
    released by an entry only the leader has.  Pinned on the reference itself."""
    tr = T.steady_trace(3, 300, 100, 4, 1, log_len=1 << 14, prune_bytes=2 << 10, name="park_commit_at_wrap")
    k = 196          # after this round end = 16356 = len - 28
    return _with_events(tr, {k: [("QUIESCE",), ("HOLD", 1), ("HOLD", 2)],
                             k + 5: [("QUIESCE",), ("RELEASE", 1), ("RELEASE", 2), ("QUIESCE",)]})


def diverge_failover():
    """BASELINE config 5 with REAL divergence at the crash: the leader loses its majority (2, 3, 4
    held), keeps appending and pushing to the one follower it still reaches -- entries that cannot
    commit -- and dies while that follower (1) is cut off as well.  2 wins the next term with the
    votes of 3 and 4.  What the reference does with 1 (pinned, tests/test_oracle_vs_refloops.py):
    both vote requests to it failed, so check_failure_count (dare_server.c:1189-1230) removes it from
    the configuration together with the dead leader, in the new leader's first pass; when 1 comes
    back nobody replicates to it any more -- it keeps its divergent log until it re-joins.  (Had 1
    been reachable it would have refused its vote -- its log is longer, poll_vote_requests
    :1661-1673 -- raised its term to the candidate's and then never followed the new leader:
    hb_receive_cb :903-910 only calls server_to_follower for a NEW term.  log_adjustment's truncation,
    dare_ibv_rc.c:1292-1451, needs a voter whose extra entries are of an OLDER term than the
    candidate's last entry, i.e. two fail-overs in a row.)"""
    tr = T.steady_trace(5, 900, (64, 107), 8, 16, log_len=1 << 18, name="diverge_failover")
    ev, k = [], 0
    for e in tr.events:
        if e[0] == "PRUNE":
            continue
        ev.append(e)
        if e[0] == "ROUND":
            k += 1
            if k == 20:
                ev += [("QUIESCE",), ("HOLD", 2), ("HOLD", 3), ("HOLD", 4)]
            if k == 26:
                ev += [("HOLD", 1), ("KILL", 0), ("RELEASE", 2), ("RELEASE", 3), ("RELEASE", 4), ("ELECT", 2), ("QUIESCE",)]
            if k == 30:
                ev += [("RELEASE", 1), ("QUIESCE",)]
    tr.events = ev
    return tr


def double_failover_truncate():
    """log_adjustment's truncation (dare_ibv_rc.c:1292-1451: not-committed buffer, dare_log.h:339;
    log_find_remote_end_offset, :367), reached the only way the reference reaches it: two fail-overs.
    Term 2: the leader loses its majority and pushes entries X to server 1 only; it dies.  Server 2
    wins term 4 with the votes of 3 and 4; server 1 -- reachable, but its log is longer -- refuses,
    raises its term to the candidate's and is left alone by the new leader (no vote ACK, hb_receive_cb
    :903-910), which commits entries Y with 3 and 4.  Then 2 dies: server 3 wins term 6, and this time
    1 votes (its last entry is of term 2, the candidate's of term 4): the new leader compares 1's
    not-committed entries with its own log, sets 1's end to the first offset that differs and
    replicates from there -- X is gone, 1 is a follower like the others."""
    tr = T.steady_trace(5, 1200, (64, 107), 8, 16, log_len=1 << 18, name="double_failover_truncate")
    ev, k = [], 0
    for e in tr.events:
        if e[0] == "PRUNE":
            continue
        ev.append(e)
        if e[0] == "ROUND":
            k += 1
            if k == 20:
                ev += [("QUIESCE",), ("HOLD", 2), ("HOLD", 3), ("HOLD", 4)]
            if k == 26:
                ev += [("KILL", 0), ("RELEASE", 2), ("RELEASE", 3), ("RELEASE", 4), ("ELECT", 2), ("QUIESCE",)]
            if k == 38:
                ev += [("QUIESCE",), ("KILL", 2), ("ELECT", 3), ("QUIESCE",)]
    tr.events = ev
    return tr


def c5_rejoin():
    """BASELINE config 5 with its optional tail: after the two kills a new server joins into the lowest
    empty slot (the dead leader's, slot 0), recovers snapshot offset and log from the group
    (rc_recover_sm / rc_recover_log, dare_ibv_rc.c:597-866), is adjusted by the leader and takes part in a
    fourth phase.  No prune tick before the join: the reference's joiner only survives its first persist
    pass when head is 0 or everything is 64-byte aligned (oracle/apus_oracle.c, orc_join -8)."""
    return T.config_c5(per_phase=600, log_len=1 << 19, batch=16, rejoin=True, prune_bytes=1 << 30)


def join_empty_slot():
    """a follower dies and is removed; later a new machine joins into its slot (Case 3 of
    handle_server_join_request, dare_ibv_ud.c:1022-1041: the bit is turned on again, STABLE stays) while
    the ring has been pruned (head > 0, 128-byte entries keep the joiner's persist walk aligned)"""
    tr = T.steady_trace(3, 900, 64, 4, 10, log_len=1 << 16, name="join_empty_slot")
    return _with_events(tr, {9: [("QUIESCE",), ("KILL", 2), ("QUIESCE",)], 40: [("QUIESCE",), ("JOIN", 2), ("QUIESCE",)]})


def join_upsize_3_to_5():
    """the group is full: two joins extend it 3 -> 4 -> 5, each through the three CONFIG entries
    EXTENDED -> TRANSIT -> STABLE (Case 4, dare_ibv_ud.c:1026-1041; apply_committed_entries
    dare_server.c:1883-1937; TRANSIT commits with the NEW group's majority, dare_ibv_rc.c:1650-1758).  Prune
    ticks between the joins commit a <HEAD> entry, which is what lets a follower dump its state machine a
    second time (dare_server.c:611, :2171)."""
    tr = T.steady_trace(3, 1500, 64, 4, 10, log_len=1 << 16, name="join_upsize_3_to_5")
    return _with_events(tr, {20: [("QUIESCE",), ("JOIN", 3), ("QUIESCE",)], 80: [("QUIESCE",), ("JOIN", 4), ("QUIESCE",)]})


def join_wrapped():
    """the joiner arrives while the log wraps (end < head): rc_recover_log fetches [head, len) only and leaves
    end = commit = 0 (dare_ibv_rc.c:806-818), the leader's log update brings the new lap"""
    tr = T.steady_trace(3, 1500, 64, 4, 10, log_len=1 << 16, name="join_wrapped")
    return _with_events(tr, {9: [("QUIESCE",), ("KILL", 1), ("QUIESCE",)], 52: [("QUIESCE",), ("JOIN", 1), ("QUIESCE",)]})


def join_then_failover():
    """a joined server is a full member: after 3 -> 4 the leader dies, the JOINED server wins the next term
    (votes, log adjustment and the blank CONFIG entry with the group of four) and leads"""
    tr = T.steady_trace(3, 900, 64, 4, 10, log_len=1 << 16, name="join_then_failover")
    return _with_events(tr, {20: [("QUIESCE",), ("JOIN", 3), ("QUIESCE",)],
                             50: [("QUIESCE",), ("KILL", 0), ("ELECT", 3), ("QUIESCE",)]})


CATALOGUE = {f.__name__: f for f in (wrap_quirk_second_round, evict_slow_follower, c5_rejoin, join_empty_slot, join_wrapped, join_upsize_3_to_5, join_then_failover, diverge_failover, double_failover_truncate, steady3, steady5_unaligned, steady7_mixed, c2_small, c3_small, c4_small,
                                     c5_failover, hold_one_of_three, hold_release, no_quorum, no_quorum_prune,
                                     exact_fit, kill_follower, park_commit_at_wrap)}


def no_quorum_wide():
    """no_quorum in a ring twice as big: stays below the 75 % fill at which the reference's
    force_log_pruning (dare_server.c:2069) evicts the slow followers -- the engine has no
    eviction, so its tests keep clear of it"""
    tr = T.steady_trace(3, 400, 64, 4, 16, log_len=1 << 17, name="no_quorum_wide")
    return _with_events(tr, {5: [("HOLD", 1), ("HOLD", 2)], 12: [("RELEASE", 1), ("QUIESCE",)]}, drop_prune=True)


# traces for the GPU tests only (not part of the golden records written from the reference)
EXTRA = {f.__name__: f for f in (no_quorum_wide,)}


def random_hold_release(seed: int, oracle_run=None):
    """Seeded random workload: group size, ring size, entry sizes and round sizes drawn per seed; followers cut off and
    released, QUIESCE events and prune ticks at random places (the majority always stays reachable).  A follower that is
    cut off for long holds the head back; at 75 % fill the reference evicts it (force_log_pruning, dare_server.c:2069 --
    modelled by the oracle only): the ring is then quadrupled until the trace stays clear of it.
    oracle_run(trace) -> cluster (oracle.oracle.run_trace) decides that; the events do not depend on the ring size."""
    import numpy as np
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([3, 5, 7]))
    L = int(rng.choice([1 << 18, 1 << 20, 1 << 22]))
    sizes = [(64,), (64, 107), (40, 64, 300, 1024), (1, 17, 64)][int(rng.integers(0, 4))]
    batch = [64, (1, 64), 16, (8, 32)][int(rng.integers(0, 4))]
    n_send = int(rng.integers(3000, 30000))
    picks = rng.random(4 * n_send + 64)          # the same decisions for every ring size tried below
    who = rng.integers(1, n, 4 * n_send + 64)
    while True:
        base = _random_events(n, n_send, sizes, batch, L, seed, picks, who)
        if oracle_run is None or oracle_run(base).force_prunes == 0 or L >= (1 << 26):
            return base
        L *= 4


def _random_events(n, n_send, sizes, batch, L, seed, picks, who):
    base = T.steady_trace(n, n_send, sizes, 8, batch, log_len=L, prune_bytes=max(L // 8, 4096), seed=seed)
    ev, held, k = [], [], 0
    for e in base.events:
        k += 1
        ev.append(e)
        p = picks[2 * (k % (len(picks) // 2))]
        if e[0] == "ROUND" and p < 0.01:
            if held and picks[2 * (k % (len(picks) // 2)) + 1] < 0.6:
                ev += [("RELEASE", held.pop()), ("QUIESCE",)]
            elif len(held) < (n - 1) // 2:                      # the majority stays reachable
                f = int(who[k % len(who)])
                if f not in held:
                    held.append(f); ev.append(("HOLD", f))
        elif e[0] == "ROUND" and p > 0.996:
            ev.append(("QUIESCE",))
    for f in held:
        ev.append(("RELEASE", f))
    ev.append(("QUIESCE",))
    base.events = ev
    return base
