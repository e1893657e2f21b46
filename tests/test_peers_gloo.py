"""CPU, gloo, world size 3 and 5: the host side of the peer-mapped group (apus_amd/peers.py)
with a stand-in for the device engine.  Checked: the handle exchange (every rank imports exactly
the blob every other rank exported), that only the current leader issues data-plane calls, and
that every rank's view of the control plane (term, configuration bitmask, who answers, leader)
follows the oracle's through elections, a leader fail-over and a follower removal (BASELINE
config 5) and through JOINs -- spare ranks that join later and extend the group 3 -> 4 -> 5, the rank of a
killed server that comes back as a new machine: only the leader asks the device to carry the join out,
only the joiner clears its own replica, every rank ends up with the leader's bitmask, size and epoch.
The device side of the same walk is tests/test_gpu_peers.py."""
import ctypes as C
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeLib:
    def __init__(self, eng):
        self.eng = eng

    def apus_gpu_export_replica(self, h, replica, out):
        o = out._obj
        o.replica, o.log_len, o.dir_cap, o.device = replica, self.eng.log_len, 4096, 0
        for k in range(6):
            for b in range(64):
                o.handle[k][b] = (replica * 37 + k * 7 + b) & 0xFF
        return 0

    def apus_gpu_import_replica(self, h, inp):
        o = inp._obj
        self.eng.imported[o.replica] = bytes(o)
        return 0

    def apus_gpu_unmap_peers(self, h):
        self.eng.calls.append(("unmap_peers",))
        return 0

    # the receiver's fence: a replica's ring + mailbox move (handles 0 and 6 change, `fences` counts), the others map them
    def apus_gpu_fence_replica(self, h, replica, out):
        self.eng.fence_n += 1
        self.eng.calls.append(("fence", replica, self.eng.fence_n))
        self.apus_gpu_export_replica(h, replica, out)
        o = out._obj
        o.fences = self.eng.fence_n
        for k in (0, 6):
            o.handle[k][0] = (o.handle[k][0] + self.eng.fence_n) & 0xFF
        return 0

    def apus_gpu_remap_fenced(self, h, inp):
        o = inp._obj
        assert o.replica in self.eng.imported, "remap of a replica that was never imported"
        self.eng.calls.append(("remap", o.replica, o.fences))
        self.eng.imported[o.replica] = bytes(o)
        return 0


class _FakeEngine:
    """records what the host asks of the device; control-plane bookkeeping as in apus_amd/engine.py"""

    def __init__(self, group_size, log_len, local_ids=None, device=0, flags=0, capacity=None):
        from apus_amd.engine import Engine
        self.group_size, self.log_len, self.local_ids = group_size, log_len, list(local_ids)
        self.capacity = capacity or group_size
        self.epoch, self.machines, self.snap_head, self.joined_at = 0, group_size, {}, 0
        self._join = Engine.join
        self.h = None
        self.L = _FakeLib(self)
        self.imported = {}
        self.fence_n = 0
        self.calls = []
        self.leader, self.term = -1, 0
        self.bitmask = self.reachable = (1 << group_size) - 1
        self.round_of_g0 = {}
        self._elect, self._kill = Engine.elect, Engine.kill
        self.n_passes = 0

    def _chk(self, rc, what):
        assert rc == 0, what

    def stage_trace(self, tr):
        rounds = [(ev[1], ev[2]) for ev in tr.events if ev[0] == "ROUND"]
        self.round_of_g0 = {g0: i for i, (g0, _) in enumerate(rounds)}

    def elect(self, w): self._elect(self, w)
    def join(self, r): self._join(self, r)
    def offsets(self, r): return {"head": self.n_passes}          # (moves with every pass: a <HEAD> entry was committed)
    def kill(self, r): self._kill(self, r)
    def set_reachable(self, mask): self.reachable = mask
    def hold(self, r): self.set_reachable(self.reachable & ~(1 << r))
    def release(self, r): self.set_reachable(self.reachable | (1 << r))
    def append_control(self, t, data=None): self.calls.append(("control", t))
    def _cid_bytes(self): return b"\0" * 16
    def run_rounds(self, r0, n): self.calls.append(("rounds", r0, n)); self.n_passes += n
    def tick_prune(self): self.calls.append(("prune",))
    def quiesce(self): self.calls.append(("quiesce",))
    def sync(self): self.calls.append(("sync",))
    def round_record(self): return [0] * self.n_passes, [0] * self.n_passes
    def close(self): pass

    # the replica kernels (PeerMember.rep_begin / rep_end)
    def rep_start(self, idle_ms=0, peer_ms=0, n_append=0, n_fwork=0): self.calls.append(("rep_start", self.c_leader))
    def rep_park(self): self.calls.append(("rep_park",)); return 0
    def rep_drain(self, timeout_ms=0): self.calls.append(("rep_drain",))
    def rep_run(self, r0, n): self.calls.append(("rep_run", r0, n)); self.n_passes += n
    def rep_prune(self): self.calls.append(("rep_prune",))
    def status_names(self): return "OK"
    c_leader = -1                       # what the C engine was told (apus_gpu_set_leader / an election it ran itself)

    # Engine.elect calls the C ABI through these two
    class _L:
        pass


def _worker(rank, world, port, name, q, replica=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from apus_amd import _lib, peers
        from oracle import oracle as orc
        from tests import traces

        tr = traces.CATALOGUE[name]()
        assert tr.group_size <= world          # the other ranks are machines that JOIN later

        def factory(n, log_len, local_ids=None, device=0, flags=0, capacity=None):
            e = _FakeEngine(n, log_len, local_ids, device, flags, capacity)

            def fake_join(h, r, lid, bitmask, reachable, out):
                # what apus_gpu_join reports back: the new bitmask, group size and epoch
                up = r == e.group_size
                out[0], out[1], out[2] = bitmask | (1 << r), e.group_size + (1 if up else 0), e.epoch + (1 if up else 0)
                e.calls.append(("join", r, lid))
                return 0
            e.L.apus_gpu_join = fake_join
            e.L.apus_gpu_clear_replica = lambda h, r: (e.calls.append(("clear", r)), 0)[1]
            e.L.apus_gpu_set_config = lambda h, n_, ep: (e.calls.append(("config", n_, ep)), 0)[1]
            e.L.apus_gpu_become_leader_ex = lambda h, w, term, bm, dead: (e.calls.append(("lead", w, term, bm, dead)), 0)[1]
            e.L.apus_gpu_set_reachable = lambda h, m: 0

            def fake_set_leader(h, l):
                e.c_leader = l
                e.calls.append(("set_leader", l))
                return 0
            e.L.apus_gpu_set_leader = fake_set_leader

            def fake_elect(h, w, live, bm, out):
                out[0], out[1], out[2], out[4] = 1, live & bm & ~(1 << w), 0, bin(live & bm).count('1')
                return 0
            e.L.apus_gpu_elect = fake_elect
            return e

        m = peers.PeerMember(world, rank, 0, tr.log_len, engine_factory=factory, configured=tr.group_size)
        # the exchange: everybody else's blob, byte for byte
        for r in range(world):
            if r == rank:
                assert r not in m.eng.imported
                continue
            want = _lib.IpcReplica()
            m.eng.L.apus_gpu_export_replica(None, r, C.byref(want))
            assert m.eng.imported[r] == bytes(want), f"rank {rank}: blob of replica {r} arrived damaged"

        cl = orc.Cluster(tr.group_size, tr.log_len)
        reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
        pos = 0
        led_calls = {}

        def check(i, ev, mm):
            nonlocal pos
            while pos <= i and pos < len(tr.events):
                e = tr.events[pos]
                if e[0] == "ROUND":
                    cl.round(reqs[e[1]:e[1] + e[2]], tr.arena)
                else:
                    getattr(cl, {"ELECT": "elect", "KILL": "kill", "PRUNE": "tick_prune", "QUIESCE": "quiesce",
                                 "HOLD": "hold", "RELEASE": "release", "JOIN": "join"}[e[0]])(*e[1:])
                pos += 1
            assert mm.leader == cl.leader, f"rank {rank} event {i}: leader {mm.leader} vs {cl.leader}"
            if cl.leader >= 0:
                sid = cl.sid(cl.leader)
                assert mm.eng.term == sid >> 9, f"rank {rank} event {i}: term {mm.eng.term} vs {sid >> 9}"
                assert mm.eng.bitmask == cl.cid_bitmask(cl.leader), f"rank {rank} event {i}: configuration"
                cid = cl.cid(cl.leader)
                assert (mm.eng.group_size, mm.eng.epoch) == (cid["size0"], cid["epoch"]), f"rank {rank} event {i}: size / epoch"
            data = [c for c in mm.eng.calls if c[0] in ("rounds", "prune", "quiesce", "control", "lead")]
            led_calls[i] = (mm.is_leader, len(data))

        def check_fences(mm):
            # every election but the first: this rank's replica left its ring + mailbox once, and it mapped what every other
            # rank moved -- with that rank's count, in the same election
            n_el = sum(1 for e in tr.events if e[0] == "ELECT")
            fences = [c for c in mm.eng.calls if c[0] == "fence"]
            assert [c[1:] for c in fences] == [(rank, k + 1) for k in range(n_el - 1)], f"rank {rank}: {fences} for {n_el} elections"
            remaps = [c for c in mm.eng.calls if c[0] == "remap"]
            assert sorted(remaps) == sorted(("remap", r, k + 1) for k in range(n_el - 1) for r in range(world) if r != rank), f"rank {rank}: {remaps}"
            assert mm.fenced == n_el - 1

        if replica:
            peers.walk_trace(m, tr, on_check=check, check_at=("QUIESCE", "ELECT", "KILL", "JOIN"), replica=True)
            check_fences(m)
            calls = m.eng.calls
            starts = [k for k, c in enumerate(calls) if c[0] == "rep_start"]
            parks = [k for k, c in enumerate(calls) if c[0] == "rep_park"]
            # every run this process took part in was parked again, in order
            assert len(starts) == len(parks) and all(a < b for a, b in zip(starts, parks)), f"rank {rank}: {calls}"
            assert all(b < a2 for b, a2 in zip(parks, starts[1:]))
            led_terms = 0
            for a, b in zip(starts, parks):
                inside = [c[0] for c in calls[a + 1:b]]
                if any(c in ("rep_run", "rep_prune", "rep_drain") for c in inside):
                    led_terms += 1              # only a process that leads feeds the run and drains it
                else:
                    # a follower's process: told who leads right before it launched its own replica's workgroups
                    assert calls[a - 1][0] == "set_leader" and calls[a - 1][1] == calls[a][1] != rank, f"rank {rank}: {calls[a - 2:a + 1]}"
            if not m.led:
                assert led_terms == 0 and not any(c[0] in ("rep_run", "rep_prune") for c in calls)
            q.put((rank, True, len(m.led), sum(c[2] for c in calls if c[0] == "rep_run"), len(starts)))
            m.close()
            return
        peers.walk_trace(m, tr, on_check=check, check_at=("QUIESCE", "PRUNE", "ELECT", "KILL", "JOIN"))
        check_fences(m)
        data = [c for c in m.eng.calls if c[0] in ("rounds", "prune", "quiesce", "control", "join")]
        joins = [e[1] for e in tr.events if e[0] == "JOIN"]
        # a machine that joins clears the replica it hosts; nobody else's is cleared from here
        assert [c[1] for c in m.eng.calls if c[0] == "clear"] == [r for r in joins if r == rank]
        if not m.led:
            assert not data, f"rank {rank} never led but issued {data[:3]}"
        else:
            assert any(c[0] == "rounds" for c in data)
        q.put((rank, True, len(m.led), sum(c[2] for c in data if c[0] == "rounds")))
        m.close()
    except BaseException as exc:      # noqa: BLE001
        import traceback
        q.put((rank, False, repr(exc) + traceback.format_exc()[-1200:], 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("steady3", 3), ("c5_failover", 5), ("hold_release", 5),
                                        ("join_upsize_3_to_5", 5), ("c5_rejoin", 5)])
def test_peer_group_host_logic(name, world):
    from oracle import oracle as orc
    if not hasattr(orc.Cluster, "cid_bitmask"):
        pytest.skip("oracle without cid accessor")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for r in sorted(res):
        assert r[1], f"rank {r[0]}: {r[2]}"
    from tests import traces
    tr = traces.CATALOGUE[name]()
    n_rounds = sum(1 for e in tr.events if e[0] == "ROUND")
    assert sum(r[3] for r in res) == n_rounds            # every round was run by exactly one rank
    if name == "c5_failover":
        assert [r[2] for r in sorted(res)] == [1, 1, 0, 0, 0]


@pytest.mark.parametrize("name,world", [("steady3", 3), ("c5_failover", 5), ("hold_release", 5), ("c5_rejoin", 5)])
def test_peer_group_replica_kernels_host_logic(name, world):
    """The replica-kernel walk (every process runs the workgroups of the replica it hosts): who starts a run,
    who is told the leader first, who feeds and drains it, that every run is parked before a control-plane
    event, that a killed / held server's process stays out and a joined one comes in."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, name, q, True)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for r in sorted(res):
        assert r[1], f"rank {r[0]}: {r[2]}"
    from tests import traces
    tr = traces.CATALOGUE[name]()
    n_rounds = sum(1 for e in tr.events if e[0] == "ROUND")
    assert sum(r[3] for r in res) == n_rounds            # every round went through exactly one leader's run
    runs = {r[0]: r[4] for r in res}
    if name == "c5_failover":
        # rank 0 led and was killed: it takes part in fewer runs than the servers that stay
        assert runs[0] < runs[1] and [r[2] for r in sorted(res)] == [1, 1, 0, 0, 0]
    if name == "hold_release":
        held = [e[1] for e in tr.events if e[0] == "HOLD"]
        assert all(runs[h] < max(runs.values()) for h in held)      # a held server's process sits runs out


def _deposed_worker(rank, world, port, q):
    """host logic of tests/test_gpu_peers_deposed.py (stand-in engine): ranks 1 and 2 hold an election among themselves, rank 0 -- the
    deposed leader behind a partition -- is not part of it"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from apus_amd import _lib, peers

        def factory(n, log_len, local_ids=None, device=0, flags=0, capacity=None):
            e = _FakeEngine(n, log_len, local_ids=local_ids, device=device, flags=flags, capacity=capacity)
            e._elect = lambda self, w: (setattr(self, "term", getattr(self, "term", 0) + 2), setattr(self, "leader", w), self.calls.append(("elect", w)))
            return e
        sub = dist.new_group(ranks=[1, 2])
        m = peers.PeerMember(world, rank, 0, 1 << 16, engine_factory=factory)
        first = {r: m.eng.imported[r] for r in m.eng.imported}
        m.elect(0)                                           # a group's first election: nobody to fence off
        assert m.fenced == 0 and not any(c[0] in ("fence", "remap") for c in m.eng.calls)
        dist.barrier()
        if rank != 0:
            m.pg = sub
            m.kill(0)
            m.elect(1)                                       # the survivors' election: both leave their ring + mailbox, map each other's
            other = 3 - rank
            assert m.fenced == 1 and ("fence", rank, 1) in m.eng.calls and ("remap", other, 1) in m.eng.calls
            assert not any(c[0] == "remap" and c[1] == 0 for c in m.eng.calls)          # nothing of the deposed leader's moved
            assert m.eng.imported[other] != first[other] and _lib.IpcReplica.from_buffer_copy(m.eng.imported[other]).fences == 1
            assert m.leader == 1
        else:
            # the deposed leader took no part: what it has mapped of ranks 1 and 2 is what they have LEFT
            assert m.fenced == 0 and all(m.eng.imported[r] == first[r] for r in (1, 2)) and m.leader == 0
        dist.barrier()
        q.put((rank, True, ""))
    except BaseException as exc:      # noqa: BLE001
        import traceback
        q.put((rank, False, repr(exc) + traceback.format_exc()[-1200:]))
    finally:
        dist.destroy_process_group()


def test_a_deposed_leader_keeps_the_mappings_the_others_left():
    """the receiver's fence at the level of the host logic (CPU, gloo, stand-in engine): an election on a sub-group moves and
    re-maps only its members' buffers; the rank outside keeps handles of allocations nobody reads any more"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_deposed_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for r in sorted(res):
        assert r[1], f"rank {r[0]}: {r[2]}"
