"""One replica per GPU / process with peer-mapped logs: the data plane of `bench.py --gpus N`.

The reference's followers are passive while the leader replicates: its NIC writes their log,
`end` and `commit` words in place (src/dare/dare_ibv_rc.c:1465-1643, 1761-1819) once the
RC_SYN / RC_SYNACK handshake has exchanged MR addresses and rkeys (dare_ibv_ud.c:1098-1380).
Here rank r hosts replica r in its own GPU's HBM, exports its six buffers as HIP IPC handles
(apus_gpu_export_replica) and maps everybody else's (apus_gpu_import_replica) -- that exchange,
once at start-up, is the only thing torch.distributed carries.  From then on the engine of
whichever rank leads runs the SAME fused kernels as for logical replicas on one device; their
stores to the followers' rings, directories, control blocks and apply streams are peer stores
(xGMI between GPUs).  No message, host read-back or host round trip per batch: the followers'
processes read their own HBM when they want to look (tests: at every quiescent event), and any
rank can take over as leader from what is in the control blocks (k_set_roles).

Ranks walk one trace together (SPMD).  Only control-plane events synchronise them: ELECT and
KILL(leader) (a barrier, the election's message exchange in the reference) and the test's check
points.  ROUND / PRUNE / QUIESCE are carried out by the leader alone."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .engine import Engine, EngineError
from .trace import DEFAULT_LOG


def _gather_bytes(blob: bytes, device, group=None) -> list[bytes]:
    """all_gather of equal-sized byte strings (device tensors on RCCL, host tensors on gloo)."""
    world = dist.get_world_size(group=group)
    on_dev = dist.get_backend() == "nccl"
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    if on_dev:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return [bytes(o.cpu().numpy().tobytes()) for o in outs]


class PeerMappingUnavailable(EngineError):
    """some rank could not map a peer's buffers (no IPC / peer access between the devices)"""


class PeerMember:
    """The replica this rank hosts + the mapped replicas of every peer."""

    def __init__(self, group_size: int, rank: int, device_index: int, log_len: int = DEFAULT_LOG, flags: int = 0,
                 engine_factory=Engine, configured: int | None = None):
        """group_size = processes = replicas that exist; configured (<= group_size) = servers of the initial
        configuration -- the other ranks are machines that JOIN later (apus_gpu_join extends the group)."""
        if dist.get_world_size() != group_size:
            raise EngineError("one process per replica: world size must equal the number of replicas")
        self.n, self.rank = group_size, rank
        self.device = torch.device("cuda", device_index)
        kw = {} if configured is None or configured == group_size else {"capacity": group_size}
        n0 = group_size if not kw else configured
        for attempt in range(4):
            # several processes opening the same device at the same moment: engine creation has been seen to
            # fail once in a while on a fresh box (one rank of five); it is retried before the group gives up
            try:
                self.eng = engine_factory(n0, log_len, local_ids=[rank], device=device_index, flags=flags, **kw)
                break
            except EngineError:
                if attempt == 3:
                    raise
                import time
                time.sleep(0.25 * (attempt + 1))
        self.log_len = log_len
        L = self.eng.L
        mine = _lib.IpcReplica()
        rc_exp = L.apus_gpu_export_replica(self.eng.h, rank, C.byref(mine))
        blobs = _gather_bytes(bytes(mine), self.device)
        failed = f"export_replica rc={rc_exp}" if rc_exp else None
        for r, b in enumerate(blobs):
            if r == rank or failed:
                continue
            h = _lib.IpcReplica.from_buffer_copy(b)
            rc = L.apus_gpu_import_replica(self.eng.h, C.byref(h)) if h.replica == r else -1
            if rc:
                failed = f"import_replica({r}) rc={rc}"
        # collective verdict (also: every mapping exists before anybody leads)
        ok = torch.tensor([0 if failed else 1], dtype=torch.int32, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            self.eng.close()
            raise PeerMappingUnavailable(failed or "a peer could not map this group's buffers")
        self.eng.local_ids = list(range(group_size))
        self.leader = -1
        self.led = []                       # (first pass, passes) of the round record for every term this rank led
        self.rep_running = False            # a run of the replica kernels is resident (rep_begin .. rep_end)
        self.rep_here = False               # ... and this process carries workgroups of it
        self.fence_on_elect = not os.environ.get("APUS_PEER_NO_RING_FENCE")     # (diagnostic: tests/test_gpu_peers_deposed.py shows what happens without it)
        self.had_leader = False
        self.fenced = 0
        self.pg = None                      # the process group the control-plane barriers run on (None: everybody; the
                                            # survivors' group once a rank has DIED -- tests/_peer_kill_worker.py)

    def _barrier(self):
        dist.barrier(group=self.pg)

    @property
    def is_leader(self) -> bool:
        return self.leader == self.rank

    # ---- control plane: every rank keeps the same view (term, configuration, who answers) ----------
    def elect(self, winner: int):
        """ELECT(winner).  The barrier stands for the vote exchange; the previous leader's stream
        has drained before it (kill / the end of its last call), so the control blocks the winner
        starts from (k_set_roles) are final.
        The receiver's fence (rc_revoke_log_access, dare_ibv_rc.c:2156-2243): every server that takes part adopts the
        new term by LEAVING the log ring and the mailbox the old leader has mapped (apus_gpu_fence_replica), and the
        members of the new term map the new ones -- a deposed leader that was not part of the election (a partition; its
        resident kernel may still be pushing) stores into memory nobody reads from then on.  The first election of a
        group has nobody to fence off."""
        e = self.eng
        if self.is_leader:
            e.sync()
        self._barrier()
        if self.fence_on_elect and (self.leader >= 0 or self.had_leader):
            self._fence_and_remap()
        if self.rank == winner:
            first = len(e.round_record()[0])
            e.elect(winner)                 # term += 2, blank CONFIG (+ removal of the dead) on the device
            self.led.append(first)
        else:
            if not (e.reachable >> winner) & 1:
                raise EngineError("the winner of an election must be alive")
            e.term += 2
            e.leader = winner
            e.bitmask &= ~(e.bitmask & ~e.reachable & ~(1 << winner))
        self.leader = winner
        self.had_leader = True

    def _fence_and_remap(self):
        """every rank of the control plane's group (self.pg) moves its replica's ring + mailbox, the handles go round, everybody
        maps what the others moved.  Collective over self.pg; ranks outside it keep their (now stale) mappings."""
        e, L = self.eng, self.eng.L
        mine = _lib.IpcReplica()
        e._chk(L.apus_gpu_fence_replica(e.h, self.rank, C.byref(mine)), "fence_replica")
        blobs = _gather_bytes(bytes(mine), self.device, group=self.pg)
        for b in blobs:
            h = _lib.IpcReplica.from_buffer_copy(b)
            if h.replica != self.rank:
                e._chk(L.apus_gpu_remap_fenced(e.h, C.byref(h)), f"remap_fenced({h.replica})")
        self.fenced += 1
        self._barrier()                      # every mapping of the new term exists before anybody leads

    def kill(self, r: int):
        e = self.eng
        if self.is_leader and r != self.rank:
            e.kill(r)                       # CONFIG entry that removes it
            return
        if r == self.leader:
            if self.is_leader:
                e.sync()                    # what it had posted is on the wire; nothing more comes
            self.leader = -1
            e.leader = -1
        elif (e.bitmask >> r) & 1 and self.leader >= 0:
            e.bitmask &= ~(1 << r)
        e.set_reachable(e.reachable & ~(1 << r))

    def hold(self, r: int): self.eng.hold(r)
    def release(self, r: int): self.eng.release(r)

    def join(self, r: int):
        """JOIN(r): the process of slot r is a NEW machine -- it zeroes the replica it hosts (log_new); the
        leader's engine does the rest on the device through the mappings (apus_gpu_join: CONFIG entries,
        the joiner's recovery as a bulk transfer into ITS HBM, first persist / apply passes); every other
        rank takes over the configuration the leader made.  Barriers = the JOIN request / reply exchange."""
        e = self.eng
        if self.is_leader:
            e.sync()
        if self.rank == r:
            e._chk(e.L.apus_gpu_clear_replica(e.h, r), "clear_replica")
        self._barrier()
        err = None
        if self.is_leader:
            try:
                e.join(r)
                e.sync()
                cfg = torch.tensor([e.bitmask, e.group_size, e.epoch, e.machines, 1], dtype=torch.int64)
            except EngineError as exc:      # (the other ranks wait in the broadcast: they must hear about it)
                err = exc
                # a refusal behind the CONFIG entries (APUS_E_NOANSWER): the leader's engine has adopted the configuration the
                # device holds -- the other ranks adopt it too (flag 2), THEN everybody raises
                cfg = torch.tensor([e.bitmask, e.group_size, e.epoch, e.machines, 2 if getattr(e, "join_refused_with_config", False) else 0], dtype=torch.int64)
                e.join_refused_with_config = False
        else:
            cfg = torch.zeros(5, dtype=torch.int64)
        if dist.get_backend() == "nccl":
            cfg = cfg.to(self.device)
        dist.broadcast(cfg, src=self.leader, group=self.pg)
        if int(cfg.cpu()[4]) != 1:
            if int(cfg.cpu()[4]) == 2 and not self.is_leader:
                bitmask, size, epoch, machines = (int(v) for v in cfg.cpu().tolist()[:4])
                e._chk(e.L.apus_gpu_set_config(e.h, size, epoch), "set_config")
                e.bitmask, e.group_size, e.epoch, e.machines = bitmask, size, epoch, machines
            raise err if err is not None else EngineError(f"rank {self.rank}: the leader could not carry out JOIN({r})")
        if not self.is_leader:
            bitmask, size, epoch, machines = (int(v) for v in cfg.cpu().tolist()[:4])
            e._chk(e.L.apus_gpu_set_config(e.h, size, epoch), "set_config")
            e.bitmask, e.group_size, e.epoch, e.machines = bitmask, size, epoch, machines
            e.set_reachable(e.reachable | (1 << r))

    # ---- data plane, replica kernels: EVERY process runs the workgroups of the replica it hosts --------
    def rep_begin(self, n_append: int = 0, n_fwork: int = 0, idle_ms: int = 20000, peer_ms: int = 2000):
        """Start a run of the replica kernels (apus_amd/csrc/apus_replica.h) across the group: every process that
        hosts a live follower launches that follower's workgroups on ITS device (they poll their mailbox), then
        the leader's process launches the leader's.  The leader pushes log bytes + one doorbell per round through
        the mappings; each follower persists, writes its reply bytes into the leader's log and its round ACK into
        the leader's mailbox (R3) from its own kernel; the leader commits by majority.  Collective."""
        e = self.eng
        if self.rep_running:
            return
        if self.is_leader:
            e.sync()
        self._barrier()                      # the control plane's last words are in everybody's control blocks
        alive = self.leader >= 0 and (e.reachable >> self.rank) & 1 and (e.bitmask >> self.rank) & 1
        self.rep_here = False
        if alive and not self.is_leader:
            e._chk(e.L.apus_gpu_set_leader(e.h, self.leader), "set_leader")
            e.rep_start(idle_ms, peer_ms, n_append, n_fwork)
            self.rep_here = True
        self._barrier()                      # every follower's workgroups are resident
        if self.is_leader:
            e.rep_start(idle_ms, peer_ms, n_append, n_fwork)
            self.rep_here = True
        self.rep_running = True

    def rep_end(self):
        """Park the run: the leader drains and parks (its last act tells every follower to park through its mailbox),
        the follower processes wait for their workgroups to leave.  Collective."""
        if not self.rep_running:
            return
        e = self.eng
        code = 0
        if self.rep_here:
            if self.is_leader:
                e.rep_drain(timeout_ms=60000)
            code = e.rep_park()
        self.rep_running = self.rep_here = False
        self._barrier()
        if code != 0:
            raise EngineError(f"rank {self.rank}: the replica kernels left with code {code}, status {e.status_names()}")

    def rep_rounds(self, r0: int, n: int):
        if self.is_leader:
            self.eng.rep_run(r0, n)

    def rep_prune(self):
        if self.is_leader:
            self.eng.rep_prune()

    # ---- data plane: the leader only ------------------------------------------------------------
    def rounds(self, r0: int, n: int):
        if self.is_leader:
            self.eng.run_rounds(r0, n)

    def tick_prune(self):
        if self.is_leader:
            self.eng.tick_prune()

    def quiesce(self):
        if self.is_leader:
            self.eng.quiesce()

    def settle(self):
        """Check point: the leader's stream has drained, everybody may look at its own replica."""
        if self.is_leader:
            self.eng.sync()
        self._barrier()

    def check_done(self):
        """Closes a check point: nobody moves on before EVERY rank has finished looking at its replica.  (Without it
        the leader went on with the next rounds while a slower rank was still reading its own memory -- a rank that
        read its offsets early and its counters late saw the counters of up to one stretch of rounds later:
        the intermittent `apply_count 50 vs 0` of round 3's join_upsize_3_to_5, rank 4's first and therefore
        slowest comparison, five rounds of ten entries before the next check point.)"""
        if os.environ.get("APUS_PEER_NO_CHECK_BARRIER"):        # diagnostic only: reproduces that failure (tests/test_gpu_peers.py)
            return
        self._barrier()

    def close(self):
        try:
            self.eng.sync()
        except EngineError:
            pass
        self._barrier()                      # nobody unmaps / frees while a peer may still write
        rc = self.eng.L.apus_gpu_unmap_peers(self.eng.h)
        self._barrier()                      # nobody frees what a peer still has mapped
        if rc:
            raise EngineError(f"rank {self.rank}: unmap_peers rc={rc}: a peer's buffers stay mapped here (a kernel of this group is still resident?)")
        self.eng.close()


def walk_trace(m: PeerMember, trace, on_check=None, check_at=("QUIESCE",), max_batch_rounds: int = 4096, batch: bool = False,
               replica: bool = False, rep_grid=(0, 0)):
    """Every rank walks the trace; on_check(i, event, member) runs on every rank between settle() and check_done():
    the leader's stream has drained and nothing is launched until every rank has looked.
    batch=True: stretches of ROUND / PRUNE events become one multi-segment launch (apus_gpu_batch_*).
    replica=True: stretches of ROUND / PRUNE events run through the replica kernels -- every process runs the
    workgroups of its own replica; the control-plane events park the run."""
    m.eng.stage_trace(trace)            # any rank may have to lead
    ev, i = trace.events, 0
    opened = False
    if replica:
        while i < len(ev):
            op = ev[i][0]
            if op in ("ROUND", "PRUNE") and op not in check_at:
                m.rep_begin(*rep_grid)
                if op == "PRUNE":
                    m.rep_prune()
                    i += 1
                    continue
                j = i
                while j < len(ev) and ev[j][0] == "ROUND":
                    j += 1
                m.rep_rounds(m.eng.round_of_g0[ev[i][1]], j - i)
                i = j
                continue
            m.rep_end()
            if op == "ELECT":
                m.elect(ev[i][1])
            elif op == "PRUNE":
                m.tick_prune()
            elif op == "QUIESCE":
                m.quiesce()
            elif op == "HOLD":
                m.hold(ev[i][1])
            elif op == "RELEASE":
                m.release(ev[i][1])
            elif op == "KILL":
                m.kill(ev[i][1])
            elif op == "JOIN":
                m.join(ev[i][1])
            else:
                raise EngineError(f"trace event {ev[i]} is not supported")
            if op in check_at and on_check is not None:
                m.settle()
                on_check(i, ev[i], m)
                m.check_done()
            i += 1
        m.rep_end()
        m.settle()
        return

    def b_open():
        nonlocal opened
        if batch and m.is_leader and not opened:
            m.eng.batch_begin(); opened = True

    def b_close():
        nonlocal opened
        if opened:
            m.eng.batch_end(); opened = False

    while i < len(ev):
        op = ev[i][0]
        if op == "ROUND":
            j = i
            while j < len(ev) and ev[j][0] == "ROUND" and j - i < max_batch_rounds:
                j += 1
            b_open()
            m.rounds(m.eng.round_of_g0[ev[i][1]], j - i)
            i = j
            continue
        if op == "PRUNE" and op not in check_at:
            b_open()
        else:
            b_close()
        if op == "ELECT":
            m.elect(ev[i][1])
        elif op == "PRUNE":
            m.tick_prune()
        elif op == "QUIESCE":
            m.quiesce()
        elif op == "HOLD":
            m.hold(ev[i][1])
        elif op == "RELEASE":
            m.release(ev[i][1])
        elif op == "KILL":
            m.kill(ev[i][1])
        elif op == "JOIN":
            m.join(ev[i][1])
        else:
            raise EngineError(f"trace event {ev[i]} is not supported")
        if op in check_at and on_check is not None:
            m.settle()
            on_check(i, ev[i], m)
            m.check_done()
        i += 1
    b_close()
    m.settle()


def init_process_group_from_env(gpus: int, timeout=None):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # test hooks for a one-GPU box: every rank on device 0, handles exchanged over gloo
    backend = os.environ.get("APUS_DIST_BACKEND", "nccl")
    if os.environ.get("APUS_DIST_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    kw = {} if timeout is None else {"timeout": timeout}         # collectives end with an error instead of waiting for ever
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), **kw)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local, backend
