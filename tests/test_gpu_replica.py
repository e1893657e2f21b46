"""The replica kernels (apus_amd/csrc/apus_replica.h): the leader's pipelined workgroups push only log
bytes + round doorbells, every follower's OWN workgroups build directory / apply records from the landed
bytes, persist, write the reply byte into the sender's log and the ACK byte into its map, the leader commits
by majority over the per-replica ACK maps.  The logs, offsets, apply streams and store counts must be the
oracle's, whichever way the requests came in (pinned multi-producer ring, staged device-resident rounds)."""
import numpy as np
import pytest

from apus_amd import trace as T
from tests import traces

pytestmark = pytest.mark.gpu


def run_and_compare(tr, source="pinned", drain_each=False, replicas=None, **kw):
    from apus_amd.engine import Engine
    from oracle import oracle as orc
    from tests.parity import compare_replica, compare_apply_tail
    cl = orc.run_trace(tr)
    n = max(tr.group_size, cl.n)
    eng = Engine(tr.group_size, tr.log_len, capacity=n)
    try:
        eng.run_trace_rep(tr, source=source, drain_each=drain_each, **kw)
        eng.quiesce()
        assert eng.status() == 0, eng.status_names()
        held = [r for r in range(eng.group_size) if not (eng.reachable >> r) & 1]
        for r in (range(eng.group_size) if replicas is None else replicas):
            if r in held:
                continue
            compare_replica(eng, cl, r, tag=f"replica kernels ({source})")
            compare_apply_tail(eng, cl, r)
        return eng.rep_latency_ns()
    finally:
        eng.close()


@pytest.mark.parametrize("source", ["pinned", "staged"])
def test_rep_three_replicas_64B(source):
    tr = T.steady_trace(3, 3000, 64, 8, 64, log_len=1 << 16)
    lat = run_and_compare(tr, source)
    assert len(lat) > 0 and np.median(lat) < 5e6


@pytest.mark.parametrize("source", ["pinned", "staged"])
def test_rep_mixed_sizes_five_replicas(source):
    tr = T.steady_trace(5, 1500, (40, 64, 107, 1024, 4096), 16, (1, 64), log_len=1 << 20, seed=3)
    run_and_compare(tr, source)


def test_rep_seven_replicas_mixed_rounds_one_at_a_time():
    tr = traces.steady7_mixed()
    run_and_compare(tr, "pinned", drain_each=True)


@pytest.mark.parametrize("name", ["steady5_unaligned", "exact_fit", "c2_small", "c3_small", "c4_small"])
def test_rep_catalogue_staged(name):
    run_and_compare(traces.CATALOGUE[name](), "staged")


FULL = {"c2": T.config_c2, "c3": T.config_c3, "c4": T.config_c4, "c5": lambda: T.config_c5(), "c5_join": lambda: T.config_c5(rejoin=True)}


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4", "c5", "c5_join"])
def test_rep_full_size_staged(cfg):
    """BASELINE configs[1..4] at FULL size through the replica kernels -- the kernel `north_star` describes, a resident
    kernel per replica -- bit for bit against the oracle on EVERY replica: all 8 offsets, every defined ring byte,
    canonical digest, counters, apply-stream hash + the newest 512 apply records (compare_replica + compare_apply_tail:
    what test_batched_step_at_full_size checks for k_step).  c2: 2^20 x 64 B over two laps of the 64 MiB ring with 16
    prune ticks, 3 replicas; c3: 2^18 x 1 KiB, 5 replicas, >= 4 laps; c4: 2^18 mixed 64 B .. 4 KiB in rounds of 1 .. 64,
    7 replicas; c5: 3 x 20 000 requests, the leader and a follower killed (+ the JOIN tail and a fourth phase).
    Reference loop: dare_server.c:1012-1125 (polling), dare_ibv_rc.c:1725-1758 (commit scan)."""
    run_and_compare(FULL[cfg](), "staged", idle_ms=20000, peer_ms=5000)


def test_rep_full_size_c2_host_fed():
    """configs[1] at full size with every request crossing the multi-producer request ring (the LD_PRELOAD path's
    admission), one ROUND event at a time"""
    run_and_compare(T.config_c2(), "pinned", idle_ms=20000, peer_ms=5000)


@pytest.mark.parametrize("name", ["hold_one_of_three", "hold_release", "no_quorum_prune", "kill_follower"])
def test_rep_failures(name):
    """HOLD / RELEASE / KILL park the run; the control-plane pass brings a released follower up to date; while a
    server is held the leader's workgroups push to the others only and commit by majority -- or not at all"""
    run_and_compare(traces.CATALOGUE[name](), "staged")


def test_rep_failover_and_join():
    """config 5: leader fail-over, a follower removed, and the JOIN tail -- elections, log adjustment and the
    joiner's recovery run between runs of the replica kernels; the joined server then gets its own workgroups"""
    run_and_compare(traces.CATALOGUE["c5_rejoin"](), "staged")
    run_and_compare(traces.CATALOGUE["join_then_failover"](), "pinned")


def test_rep_a_dead_follower_costs_its_ack_not_the_round():
    """one follower's workgroups are never started (its process is gone but the leader still pushes to it):
    4 of 5 acknowledge, everything commits by majority"""
    from apus_amd.engine import Engine
    from oracle import oracle as orc
    tr = T.steady_trace(5, 64 * 40, 64, 8, 64, log_len=1 << 20, prune_bytes=1 << 40)
    eng = Engine(5, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        # follower 4's workgroups do not exist: hide it from this engine's launch by marking it "not hosted"
        lib = eng.L
        import ctypes as C
        lib.apus_gpu_rep_test_skip_follower.argtypes = [C.c_void_p, C.c_uint32]
        lib.apus_gpu_rep_test_skip_follower(eng.h, 1 << 4)
        eng.rep_start(idle_ms=3000, peer_ms=50)
        n_rounds = sum(1 for e in tr.events if e[0] == "ROUND")
        eng.rep_run(0, n_rounds)
        eng.rep_drain(timeout_ms=20000)
        st = eng.rep_stats()
        code = eng.rep_park()
        assert st["highest_rec"] == len(tr.reqs), st
        o = eng.offsets(0)
        assert o["commit"] == o["end"] == o["apply"], o
        for r in (1, 2, 3):
            assert eng.offsets(r)["end"] == o["end"]
        assert eng.offsets(4)["end"] != o["end"]              # nobody persisted there
    finally:
        eng.close()


def test_rep_a_host_consumer_that_registers_during_the_first_run_holds_the_head_back():
    """apus_gpu_rep_follower_replayed is "callable before and during a run" -- a follower whose host replay lags
    tells the leader min(device apply, host replay) as applied, also when the consumer registers while the FIRST run is
    already resident (round 4 read the word once at the start: the first term published the device's count alone and the
    leader could prune and lap what the host had not carried out).  A consumer that has replayed nothing: applied_by stays 0
    in the leader's mailbox, no prune tick moves the verified head over it (the ring is protected), and once the host says
    it has caught up the leader hears it."""
    import ctypes as C
    import time
    from apus_amd.engine import Engine
    tr = T.steady_trace(3, 64 * 96, 64, 8, 64, log_len=1 << 20, prune_bytes=1 << 17)
    eng = Engine(3, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        applied0 = eng.counters(1)["n_apply"]                              # (the blank CONFIG entry of the election)
        eng.rep_start(idle_ms=5000, peer_ms=3000)
        L = eng.L
        assert L.apus_gpu_rep_follower_replayed(eng.h, 1, 0) == 0          # registers DURING the first run: nothing replayed yet
        time.sleep(0.05)                                                   # (the apply wavefront looks at the word every 256 passes)
        ev = [e for e in tr.events if e[0] in ("ROUND", "PRUNE")]
        i, ticks = 0, 0
        while i < len(ev) and ticks < 5:                                   # fewer ticks than head moves may stay unverified (8)
            if ev[i][0] == "PRUNE":
                eng.rep_prune(); ticks += 1; i += 1
                continue
            j = i
            while j < len(ev) and ev[j][0] == "ROUND":
                j += 1
            eng.rep_run(eng.round_of_g0[ev[i][1]], j - i)
            i = j
        pr = (C.c_uint64 * 4)()
        t0 = time.time()
        while time.time() - t0 < 10:
            L.apus_gpu_rep_follower_progress(eng.h, 1, pr)
            if pr[0] > 64 * 40:
                break
            time.sleep(0.001)
        assert pr[0] > 64 * 40, f"follower 1's kernel applied only {pr[0]} entry slots"
        w = (C.c_uint64 * 8)()
        # (the mailbox is uncached device memory: readable while the run is resident)
        assert L.apus_gpu_rep_box_words(eng.h, 0, 1, w) == 0
        # (what the follower said before the consumer registered stands: the count it started the run with)
        assert w[6] <= applied0, f"the leader was told follower 1 applied {w[6]} entry slots although its host has replayed none"
        assert L.apus_gpu_rep_box_words(eng.h, 0, 2, w) == 0 and w[6] > 0          # (follower 2 has no host consumer: the device count)
        # the host catches up: the leader hears it, the run drains, everything is applied everywhere
        applied = int(pr[0])
        L.apus_gpu_rep_follower_replayed(eng.h, 1, 1 << 40)
        eng.rep_drain(timeout_ms=20000)
        L.apus_gpu_rep_box_words(eng.h, 0, 1, w)
        assert w[6] >= applied
        assert eng.rep_park() == 0
        o = eng.offsets(0)
        assert o["commit"] == o["end"] == o["apply"]
        assert eng.offsets(1)["apply"] == o["end"]
    finally:
        eng.close()


@pytest.mark.parametrize("name,source", [("steady5_unaligned", "staged"), ("exact_fit", "staged"), ("c4_small", "staged"), ("c2_small", "pinned")])
def test_rep_doorbell_metadata_says_what_the_headers_say(name, source, monkeypatch):
    """Round 5: a follower no longer READS the headers that landed in its ring -- idx and term come with the doorbell, clt_id /
    type / sender as 4 bytes per entry beside it (R_BELL_META).  Verification mode (APUS_REP_DBG & 32768): the follower's
    wavefronts read the headers as well and raise a status bit where they differ -- wraps, exact fit, mixed sizes, rounds from
    the request ring.  And the old way (APUS_REP_DBG & 16384: no metadata, headers read back) still walks the same traces."""
    monkeypatch.setenv("APUS_REP_DBG", "32768")
    run_and_compare(traces.CATALOGUE[name](), source)
    monkeypatch.setenv("APUS_REP_DBG", "16384")
    run_and_compare(traces.CATALOGUE[name](), source)


# ---- round 6: the configurations nobody ran -------------------------------------------------------

@pytest.mark.parametrize("g", [1, 5, 7])
def test_rep_full_size_c2_stream_at_every_group_size(g):
    """The metric names 1, 3, 5 and 7 replicas on the 64-B stream: configs[1]'s stream at FULL size through the replica
    kernels at the three sizes test_rep_full_size_staged[c2] does not cover -- bit for bit on every replica.  One replica:
    the quorum of one (dare_ibv_rc.c:1725-1758 with size = 1: every entry has its majority the moment the leader holds it)
    and the leader-only launch (k_replica_leader).  Reference loops: dare_server.c:1012-1125, :1751-1790."""
    run_and_compare(T.config_c2(group_size=g), "staged", idle_ms=20000, peer_ms=5000)


def test_rep_full_size_c2_single_replica_host_fed():
    """... and the single replica with every request crossing the request ring (proxy_on_read's admission)"""
    run_and_compare(T.config_c2(group_size=1), "pinned", idle_ms=20000, peer_ms=5000)


@pytest.mark.parametrize("g,steps", [(3, 101), (1, 24)])
def test_rep_bench_shaped_run_is_bit_exact(g, steps):
    """What bench.py times, digested: ONE resident launch, `steps` passes over configs[1]'s stream on the same rings (101
    steps = 250 laps of the 64 MiB ring, 1616 prune ticks, 10^8 entries) -- then every replica against an oracle that
    replayed the same commands: all 8 offsets, every defined ring byte, the canonical digest, highest_rec, apply count +
    stream hash over ALL steps, store count, and the apply records of the last step's newest entries one by one.
    The loop this replaces: dare_server.c:1012-1125."""
    from apus_amd.engine import Engine
    from tests.parity import compare_replica, compare_apply_tail, oracle_replay_steps, step_commands
    tr = T.config_c2(group_size=g)
    eng = Engine(g, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        cmds = step_commands(tr, eng.round_of_g0)
        eng.rep_start(idle_ms=20000, peer_ms=5000)
        eng.rep_cmds(cmds, steps)                     # (what bench.py does: apus_gpu_rep_cmds pushes the steps' commands)
        eng.rep_drain(timeout_ms=120000)
        assert eng.rep_park() == 0
        eng.quiesce()
        assert eng.status() == 0, eng.status_names()
        cl = oracle_replay_steps(tr, steps)
        assert eng.counters(0)["highest_rec"] == steps * len(tr.reqs)
        for r in range(g):
            compare_replica(eng, cl, r, tag=f"{steps} steps in one resident launch")
            compare_apply_tail(eng, cl, r)
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_rep_random_traces(seed):
    """The seeded random workloads of test_random_traces_through_batched_launches (3 / 5 / 7 replicas, mixed entry and round
    sizes, followers cut off and released, QUIESCE events and prune ticks at random places) through the replica kernels,
    the oracle in lock step: compared at every QUIESCE."""
    from apus_amd.engine import Engine
    from oracle import oracle as orc
    from tests.parity import lockstep_rep, compare_apply_tail
    tr = traces.random_hold_release(seed, orc.run_trace)
    eng = Engine(tr.group_size, tr.log_len)
    try:
        cl = lockstep_rep(tr, eng, "staged")
        eng.quiesce()
        for r in range(tr.group_size):
            compare_apply_tail(eng, cl, r)
    finally:
        eng.close()


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_rep_full_size_doorbell_metadata_against_the_headers(cfg, monkeypatch):
    """R_BELL_META at full size: the followers build their directory / apply records from what the LEADER says about an entry
    (4 bytes beside the doorbell); here every follower wavefront reads the headers that landed as well and raises a status
    bit where the two differ (APUS_REP_DBG & 32768) -- configs[1..3] at full size, then the usual bit-exact comparison."""
    monkeypatch.setenv("APUS_REP_DBG", "32768")
    run_and_compare(FULL[cfg](), "staged", idle_ms=20000, peer_ms=5000)


def test_rep_inherited_entries_commit_only_behind_the_terms_first_entry():
    """a follower's cumulative count says how much of the log it holds in order, not in which term it came to hold
    it.  Entries below the slot of the leader's first entry of its OWN term (H_TERM_SLOT0, the blank CONFIG entry) commit
    only once the (quorum - 1)-th largest count covers that slot.  Here the slot is moved far ahead of the log (test hook):
    two of three replicas hold and acknowledge every round, and nothing commits; with the slot where become_leader put it
    the same rounds commit."""
    import ctypes as C
    from apus_amd.engine import Engine
    tr = T.steady_trace(3, 64 * 20, 64, 8, 64, log_len=1 << 20, prune_bytes=1 << 40)
    eng = Engine(3, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        L = eng.L
        L.apus_gpu_rep_test_term_slot0.argtypes = [C.c_void_p, C.c_uint64]
        w = eng.hdr_words(0)
        slot0 = int(w[37])                                       # H_TERM_SLOT0
        assert slot0 == eng.counters(0)["n_end"] - 1             # the blank CONFIG entry of the election
        c0 = eng.counters(0)["n_commit"]
        assert L.apus_gpu_rep_test_term_slot0(eng.h, 1 << 40) == 0
        n_rounds = sum(1 for e in tr.events if e[0] == "ROUND")
        eng.rep_start(idle_ms=3000, peer_ms=100)
        eng.rep_run(0, n_rounds)
        with pytest.raises(Exception):
            eng.rep_drain(timeout_ms=300)                        # every round is in every ring and acknowledged; nothing may commit
        assert eng.rep_stats()["commit_slot"] == c0
        eng.rep_park()
        L.apus_gpu_clear_status(eng.h)
        assert eng.counters(0)["n_commit"] == c0 and eng.counters(1)["n_end"] == eng.counters(0)["n_end"]
        # the term's first entry where it really is: the next run commits everything
        assert L.apus_gpu_rep_test_term_slot0(eng.h, slot0) == 0
        import time
        n_end = eng.counters(0)["n_end"]
        eng.rep_start(idle_ms=3000, peer_ms=500)
        t0 = time.time()
        while eng.rep_stats()["commit_slot"] < n_end and time.time() - t0 < 5:
            time.sleep(0.001)
        assert eng.rep_stats()["commit_slot"] == n_end           # by the followers' counts, no new round needed
        assert eng.rep_park() == 0
        eng.quiesce()                                            # (entries without a ticket are applied by the control-plane pass)
        assert eng.status() == 0, eng.status_names()
        o = eng.offsets(0)
        assert o["commit"] == o["end"] == o["apply"], o
        assert eng.counters(0)["highest_rec"] == len(tr.reqs)
    finally:
        eng.close()


def test_rep_a_replica_moves_through_its_fence_pairs_and_round_again():
    """apus_gpu_fence_replica (include/apus_gpu.h): a replica lives in pair fences % APUS_FENCE_PAIRS of its (log ring, mailbox)
    pairs; the pairs are allocated once, a fence copies the replica into the NEXT one -- which, from the fourth fence on, holds
    what the replica looked like four fences ago.  Here every replica of a group of three (the leader's too) is fenced between
    the runs of a trace, nine times each -- twice round the pairs -- and the logs, offsets and apply streams are the oracle's
    at the end; the ring a replica left one fence ago still reads as it stood then; a ring left APUS_FENCE_PAIRS fences ago is
    the live one again and is refused."""
    import ctypes as C
    from apus_amd import _lib
    base = T.steady_trace(3, 9000, (64, 107, 300), 8, (8, 64), log_len=1 << 20, prune_bytes=1 << 17, seed=11)
    ev, k = [], 0
    for e in base.events:
        ev.append(e)
        if e[0] == "ROUND":
            k += 1
            if k % 8 == 0:
                ev.append(("QUIESCE",))
    ev.append(("QUIESCE",))
    base.events = ev
    seen = {"fences": 0, "checked": 0}

    def on_event(i, event, eng):
        if event[0] != "QUIESCE":
            return
        L = eng.L
        for r in range(3):
            before = np.empty(4096, dtype=np.uint8)
            eng._chk(L.apus_gpu_read_ring(eng.h, r, 0, 4096, before.ctypes.data), "read_ring")
            out = _lib.IpcReplica()
            eng._chk(L.apus_gpu_fence_replica(eng.h, r, C.byref(out)), "fence_replica")
            seen["fences"] += 1
            assert out.fences == seen["fences"] // 3 + (1 if seen["fences"] % 3 else 0), (out.fences, seen)
            after, left = np.empty(4096, dtype=np.uint8), np.empty(4096, dtype=np.uint8)
            eng._chk(L.apus_gpu_read_ring(eng.h, r, 0, 4096, after.ctypes.data), "read_ring")
            eng._chk(L.apus_gpu_read_retired_ring(eng.h, r, 1, 0, 4096, left.ctypes.data), "read_retired_ring")
            assert (after == before).all() and (left == before).all()
            assert L.apus_gpu_read_retired_ring(eng.h, r, 4, 0, 4096, left.ctypes.data) != 0      # (that pair is lived in again)
            seen["checked"] += 1

    run_and_compare(base, "staged", on_event=on_event)
    assert seen["fences"] >= 27, seen


# ---- round 6: the lone round's path (REP_SPEC_PAY / REP_BELL16 / REP_FAST_ACK / REP_APPLY_PRE, apus_replica.h) ------------------

@pytest.mark.parametrize("g,payload,batch,drain", [
    (3, 64, 1, True),                                   # one request at a time, each waited for: proxy_on_read with one client
    (3, 64, (1, 8), True),                              # <= 64 units: the payload comes with the descriptors
    (3, 64, (1, 12), False),                            # ... back to back: rounds the retire wavefront is and is not standing at
    (5, (16, 64, 192), (1, 4), True),                   # sizes that differ inside a round: no size in the ticket
    (3, (0, 14, 48, 64, 80, 96, 97, 200), (1, 9), True),    # header-only entries, sizes that are no multiple of 16, payloads in the arena
    (3, 448, 1, True),                                  # one size, but not in the slot
    (7, 64, (1, 2), False),
    (1, 64, (1, 3), True),
])
def test_rep_rounds_that_come_one_by_one(g, payload, batch, drain, monkeypatch):
    """Round 6 took four dependent memory round trips out of a lone round's way (host submit -> highest_rec 17 -> 14 us): the
    append wavefront asks for the slots' payload together with their descriptors when the ticket names the entries' one size
    (leader_handle_submit_req + get_tailq_message + log_append_entry, src/proxy/proxy.c:108-161, dare_ibv_ud.c:780-790); the
    doorbell carries the first eight entries' clt_id / type / sender; the follower's work wavefront sends the cumulative ACK
    itself when its retire wavefront stands at its round (rc_send_entries_reply, dare_ibv_rc.c:1828-1863); the leader's applier
    holds the round's done granules until the commit comes (dare_server.c:1815-1974).  Small rounds over a small ring -- wraps,
    prune ticks, header-only entries, payloads in the arena -- with every follower wavefront comparing what the doorbell said
    with the headers that landed (APUS_REP_DBG & 32768): every replica bit for bit the oracle's."""
    monkeypatch.setenv("APUS_REP_DBG", "32768")
    tr = T.steady_trace(g, 700, payload, 8, batch, log_len=1 << 16, seed=11)
    run_and_compare(tr, "pinned", drain_each=drain)
