"""Host-side mirror of the engine C ABI (include/apus_gpu.h) plus the trace driver.

Everything that computes runs in libapus_gpu.so on the GPU; this module only
marshals arguments, keeps the host control-plane state (who is leader, which
term, who is reachable) and maps trace events onto ABI calls.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .trace import DEFAULT_LOG, REQ_DTYPE, Trace

APPLY_DTYPE = np.dtype([("slot", "<u8"), ("off", "<u8"), ("idx", "<u8"), ("len", "<u4"),
                        ("clt_id", "<u2"), ("type", "u1"), ("kind", "u1")])
OFFSET_NAMES = ("head", "apply", "commit", "end", "tail", "old_end", "old_commit", "len")
COUNTER_NAMES = ("n_end", "n_persist", "n_commit", "n_apply", "last_idx", "sid", "highest_rec", "apply_hash")
# hdr word indices (apus_device.h)
H_APPLY_COUNT, H_PREV_HEAD, H_CID_BITMASK, H_STORE_COUNT = 16, 17, 18, 21

ST_NAMES = {1: "SECOND_WRAP", 2: "LOG_FULL", 4: "TERM_FENCE", 8: "DIR_OVERRUN", 16: "SPIN_TIMEOUT", 32: "JOIN_WALK"}


class EngineError(RuntimeError):
    rc = None            # the APUS_E_* code of the C call that failed, when there was one


class Engine:
    """N logical replicas of one consensus group on one MI355X (or the local share
    of a group that spans several processes)."""

    def __init__(self, group_size: int, log_len: int = DEFAULT_LOG, local_ids=None,
                 device: int = 0, stream: int | None = None, flags: int = 0, capacity: int | None = None):
        """capacity: replicas that exist (>= group_size); a group that is meant to grow by JOINs starts
        with fewer configured servers than replicas."""
        self.L = _lib.load()
        self.group_size = group_size
        self.capacity = capacity = max(group_size, capacity or group_size)
        self.log_len = log_len
        self.local_ids = list(range(capacity)) if local_ids is None else list(local_ids)
        cfg = _lib.Cfg()
        cfg.group_size = capacity
        cfg.n_local = len(self.local_ids)
        for k, i in enumerate(self.local_ids):
            cfg.local_ids[k] = i
        cfg.log_len = log_len
        cfg.device = device
        cfg.flags = flags          # include/apus_gpu.h APUS_F_*: 2 = term fence, 4 = strict reference quirks (the parity harness's diagnostic)
        cfg.stream = stream
        h = C.c_void_p()
        rc = self.L.apus_gpu_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise EngineError(f"apus_gpu_create failed rc={rc} (a gfx950 device is required)")
        self.h = h
        self.initial_size = group_size
        if capacity != group_size:
            self._chk(self.L.apus_gpu_set_group_size(self.h, group_size), "set_group_size")
        self.leader = -1
        self.term = 0
        self.epoch = 0
        self.machines = group_size          # LIDs handed out so far (a joiner is a new machine)
        self.bitmask = (1 << group_size) - 1
        self.reachable = (1 << group_size) - 1
        self.snap_head = {}                 # follower -> its head when it last dumped its state machine
        self.round_of_g0 = {}
        self._keep = []

    @classmethod
    def from_handle(cls, handle, group_size: int, log_len: int = DEFAULT_LOG) -> "Engine":
        """Observation-only wrapper around an engine owned by somebody else (the host C layer)."""
        self = cls.__new__(cls)
        self.L = _lib.load()
        self.h = C.c_void_p(handle)
        self.owned = False
        self.group_size, self.log_len = group_size, log_len
        self.local_ids = list(range(group_size))
        self.leader, self.term = -1, 0
        self.bitmask = self.reachable = (1 << group_size) - 1
        self.round_of_g0 = {}
        return self

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "owned", True):
                self.L.apus_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            err = EngineError(f"{what} failed rc={rc} status={self.status_names()}")
            err.rc = rc
            raise err

    def sync(self):
        self._chk(self.L.apus_gpu_sync(self.h), "sync")

    def reset(self):
        if getattr(self, "initial_size", self.group_size) != self.group_size:
            self.group_size = self.initial_size
        if getattr(self, "capacity", self.group_size) != self.group_size:
            self._chk(self.L.apus_gpu_set_group_size(self.h, self.group_size), "set_group_size")
        self._chk(self.L.apus_gpu_reset(self.h), "reset")
        self.leader, self.term, self.epoch = -1, 0, 0
        self.machines = self.group_size
        self.snap_head = {}
        self.bitmask = self.reachable = (1 << self.group_size) - 1

    # -- admission --------------------------------------------------------------
    def stage(self, reqs: np.ndarray, arena: np.ndarray, round_n: np.ndarray):
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        round_n = np.ascontiguousarray(round_n, dtype=np.uint32)
        self._chk(self.L.apus_gpu_stage(self.h, reqs.ctypes.data, len(reqs), arena.ctypes.data, len(arena),
                                        round_n.ctypes.data, len(round_n)), "stage")
        self.n_rounds = len(round_n)

    def stage_trace(self, trace: Trace):
        """Stage every ROUND of a trace; returns {g0: round index}."""
        rounds = [(ev[1], ev[2]) for ev in trace.events if ev[0] == "ROUND"]
        g = 0
        for g0, n in rounds:
            assert g0 == g, "rounds must cover the request stream in order"
            g += n
        assert g == len(trace.reqs)
        self.stage(trace.reqs, trace.arena, np.array([n for _, n in rounds], dtype=np.uint32))
        self.round_of_g0 = {g0: i for i, (g0, _) in enumerate(rounds)}
        return self.round_of_g0

    # -- control plane ------------------------------------------------------------
    def elect(self, winner: int):
        """ELECT(winner): every live server becomes a candidate of term t+1, the
        winner's election timeout fires first and it wins term t+2 (the schedule of
        oracle/apus_oracle.c:orc_elect, from dare_server.c:1264-1518)."""
        if not (self.reachable >> winner) & 1:
            raise EngineError("the winner of an election must be alive")
        # the votes are cast on the device (k_elect): who grants, who refuses (a longer log), majority or not
        out = (C.c_uint64 * 8)()
        if set(range(self.group_size)) <= set(self.local_ids):
            self._chk(self.L.apus_gpu_elect(self.h, winner, self.reachable, self.bitmask, out), "elect")
        else:
            # (the message-passing transport: this process sees one replica only; the trace's word is taken)
            out[0], out[1] = 1, self.reachable & self.bitmask & ~(1 << winner)
        self.term += 2
        if not out[0]:
            raise EngineError(f"server {winner} did not get a majority ({int(out[4])} votes)")
        self.leader = winner
        self.granted, self.refused = int(out[1]), int(out[2])
        # check_failure_count (dare_server.c:1189-1230) opens the new leader's first pass: configured
        # servers that did not answer the two vote requests are removed with a second CONFIG entry
        # that commits in the same pass as the blank one
        dead = self.bitmask & ~self.reachable & ~(1 << winner)
        self._chk(self.L.apus_gpu_set_reachable(self.h, self.reachable & self.bitmask), "set_reachable")
        self._chk(self.L.apus_gpu_become_leader_ex(self.h, winner, self.term, self.bitmask, dead), "become_leader")
        self.bitmask &= ~dead

    def _cid_bytes(self) -> bytes:
        import struct
        return struct.pack("<QBBBBI", getattr(self, "epoch", 0), self.group_size, 0, 0, 0, self.bitmask)

    def join(self, r: int):
        """JOIN(r) of the trace: a new machine joins and must be given slot r (an empty one, or the group
        size: the group is extended).  Mirrors oracle/apus_oracle.c:orc_join, incl. what it refuses."""
        if self.leader < 0:
            raise EngineError("JOIN without a leader")
        size = self.group_size
        # (a configured server that does not answer, or too few members whose OWN configuration shows the joiner:
        #  apus_gpu_join refuses with APUS_E_NOANSWER = -8, where the reference's joiner retries for ever, oracle -6)
        donors = [i for i in range(size) if i not in (r, self.leader) and (self.bitmask >> i) & 1]
        for f in donors:
            # a follower answers a second state-machine request before the next committed <HEAD> entry from
            # an uninitialised pointer in the reference (dare_server.c:604-651); <HEAD> committed == head moved
            if f in self.snap_head and self.snap_head[f] == self.offsets(f)["head"]:
                raise EngineError(f"follower {f} was asked for its state machine already and no <HEAD> entry was committed since")
        out = (C.c_uint64 * 4)()
        out[1] = 0
        rc = self.L.apus_gpu_join(self.h, r, self.machines + 1, self.bitmask, self.reachable, out)
        if rc == -8 and int(out[1]):
            # APUS_E_NOANSWER behind the CONFIG entries: the log holds them (the reference's joiner would retry for ever), the
            # configuration on the device has moved on -- this object's mirrors follow it before the refusal is raised
            self.bitmask, self.group_size, self.epoch = int(out[0]), int(out[1]), int(out[2])
            self.join_refused_with_config = True
        self._chk(rc, "join")
        self.machines += 1
        self.snap_head.pop(r, None)             # a new machine: it has not dumped anything yet
        for f in donors:
            self.snap_head[f] = self.offsets(f)["head"]
        self.bitmask, self.group_size, self.epoch = int(out[0]), int(out[1]), int(out[2])
        self.reachable |= 1 << r

    def kill(self, r: int):
        """KILL(r) of the trace: the server stops answering."""
        self.set_reachable(self.reachable & ~(1 << r))
        if r == self.leader:
            self.leader = -1
        elif self.leader >= 0 and (self.bitmask >> r) & 1:
            self.bitmask &= ~(1 << r)
            self.append_control(2, self._cid_bytes())

    def set_reachable(self, mask: int):
        """Who answers.  The leader only posts to servers that are also ON in its configuration
        (CID_IS_SERVER_ON, dare_ibv_rc.c:1480): one that was removed while it was cut off stays out
        when it comes back."""
        self.reachable = mask
        self._chk(self.L.apus_gpu_set_reachable(self.h, mask & self.bitmask), "set_reachable")

    def hold(self, r: int): self.set_reachable(self.reachable & ~(1 << r))
    def release(self, r: int): self.set_reachable(self.reachable | (1 << r))

    def append_control(self, type_: int, data: bytes | None = None):
        buf = C.create_string_buffer(data, 16) if data is not None else None
        self._chk(self.L.apus_gpu_append_control(self.h, type_, C.cast(buf, C.c_void_p) if buf else None),
                  "append_control")

    def store_stream(self, r: int, first: int, n: int) -> tuple:
        """(bytes, records): what proxy_store_cmd hands to BerkeleyDB for entry slots [first, first + n)"""
        total, recs = C.c_uint64(0), C.c_uint64(0)
        cap = 32 * n + 4096
        while True:
            buf = C.create_string_buffer(cap)
            self._chk(self.L.apus_gpu_store_stream(self.h, r, first, n, buf, cap, C.byref(total), C.byref(recs)),
                      "store_stream")
            if total.value <= cap:
                return buf.raw[:total.value], int(recs.value)
            cap = total.value

    def run_rounds(self, r0: int, n: int):
        self._chk(self.L.apus_gpu_run_rounds(self.h, r0, n), "run_rounds")

    def tick_prune(self): self._chk(self.L.apus_gpu_tick_prune(self.h), "tick_prune")

    def force_prune(self) -> dict:
        """force_log_pruning (dare_server.c:2069-2122) behind a leader pass: returns what it did"""
        out = (C.c_uint64 * 4)()
        self._chk(self.L.apus_gpu_force_prune(self.h, out), "force_prune")
        if out[1] != 0xFF:
            self.bitmask = int(out[3])
        return {"full": bool(out[0]), "removed": None if out[1] == 0xFF else int(out[1]), "appended": int(out[2])}

    def batch_begin(self): self._chk(self.L.apus_gpu_batch_begin(self.h), "batch_begin")
    def batch_end(self): self._chk(self.L.apus_gpu_batch_end(self.h), "batch_end")
    def quiesce(self): self._chk(self.L.apus_gpu_quiesce(self.h), "quiesce")

    def run_trace(self, trace: Trace, on_event=None, max_batch_rounds: int = 4096, check: bool = True):
        """Drive the engine with a trace.  Consecutive ROUND events are coalesced
        into one run_rounds call (batch boundaries never change results)."""
        self.stage_trace(trace)
        i, ev = 0, trace.events
        while i < len(ev):
            op = ev[i][0]
            if op == "ROUND":
                j = i
                while j < len(ev) and ev[j][0] == "ROUND" and j - i < max_batch_rounds:
                    j += 1
                self.run_rounds(self.round_of_g0[ev[i][1]], j - i)
                last = j - 1
                i = j
            else:
                if op == "ELECT":
                    self.elect(ev[i][1])
                elif op == "PRUNE":
                    self.tick_prune()
                elif op == "QUIESCE":
                    self.quiesce()
                elif op == "HOLD":
                    self.hold(ev[i][1])
                elif op == "RELEASE":
                    self.release(ev[i][1])
                elif op == "KILL":
                    self.kill(ev[i][1])
                elif op == "JOIN":
                    self.join(ev[i][1])
                else:
                    raise EngineError(f"trace event {ev[i]} is not supported by the engine yet")
                last = i
                i += 1
            if on_event is not None:
                on_event(last, ev[last], self)
        if check:
            self.check_status()

    # -- persistent consensus kernel (live / latency path) --------------------------
    def persist_start(self, idle_ms: int = 2000, peer_ms: int = 200):
        self._chk(self.L.apus_gpu_persist_start(self.h, idle_ms, peer_ms), "persist_start")

    def persist_submit(self, reqs: np.ndarray, arena: np.ndarray):
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        self._chk(self.L.apus_gpu_persist_submit(self.h, reqs.ctypes.data, len(reqs), arena.ctypes.data, len(arena)),
                  "persist_submit")

    def persist_prune(self): self._chk(self.L.apus_gpu_persist_prune(self.h), "persist_prune")

    def persist_drain(self, timeout_ms: int = 5000):
        rc = self.L.apus_gpu_persist_drain(self.h, timeout_ms)
        if rc != 0:
            raise EngineError(f"persistent kernel did not drain rc={rc}")

    def persist_highest_rec(self) -> int: return int(self.L.apus_gpu_persist_highest_rec(self.h))

    def persist_stop(self) -> int: return int(self.L.apus_gpu_persist_stop(self.h))

    def persist_roundtrip_ns(self, reqs: np.ndarray, arena: np.ndarray, iters: int) -> np.ndarray:
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        out = np.zeros(iters, dtype=np.uint32)
        self._chk(self.L.apus_gpu_persist_roundtrip(self.h, reqs.ctypes.data, len(reqs), arena.ctypes.data,
                                                    len(arena), iters, out.ctypes.data), "persist_roundtrip")
        return out

    def persist_latency_ns(self) -> np.ndarray:
        out = np.zeros(1 << 16, dtype=np.uint32)
        n = C.c_uint32(0)
        self._chk(self.L.apus_gpu_persist_latency(self.h, out.ctypes.data, len(out), C.byref(n)), "persist_latency")
        return out[:n.value].copy()

    def persist_latency_phase_ns(self, which: int) -> np.ndarray:
        out = np.zeros(1 << 16, dtype=np.uint32)
        n = C.c_uint32(0)
        self._chk(self.L.apus_gpu_persist_latency_phase(self.h, which, out.ctypes.data, len(out), C.byref(n)),
                  "persist_latency_phase")
        return out[:n.value].copy()

    def run_trace_persistent(self, trace: Trace, idle_ms: int = 2000, peer_ms: int = 200):
        """Same events, but the ROUND / PRUNE events go through the persistent kernel's
        command ring (ELECT, HOLD/RELEASE and QUIESCE stop it and use the phased path)."""
        reqs = np.ascontiguousarray(trace.reqs, dtype=REQ_DTYPE)
        arena = np.ascontiguousarray(trace.arena, dtype=np.uint8)
        running = False

        def stop():
            nonlocal running
            if running:
                self.persist_drain()
                code = self.persist_stop()
                running = False
                if code not in (0,):
                    raise EngineError(f"persistent kernel exited with code {code}")
        try:
            for ev in trace.events:
                op = ev[0]
                if op in ("ROUND", "PRUNE"):
                    if not running:
                        self.persist_start(idle_ms, peer_ms)
                        running = True
                    if op == "ROUND":
                        self.persist_submit(reqs[ev[1]:ev[1] + ev[2]], arena)
                    else:
                        self.persist_prune()
                    continue
                stop()
                if op == "ELECT":
                    self.elect(ev[1])
                elif op == "QUIESCE":
                    self.quiesce()
                elif op == "HOLD":
                    self.hold(ev[1])
                elif op == "RELEASE":
                    self.release(ev[1])
                else:
                    raise EngineError(f"trace event {ev} is not supported")
        finally:
            if running:
                try:
                    self.persist_stop()
                except Exception:
                    pass
        self.check_status()

    # -- replica kernels: every replica runs its own resident workgroups (apus_replica.h) ----------
    def rep_start(self, idle_ms: int = 3000, peer_ms: int = 500, n_append: int = 0, n_fwork: int = 0):
        self._chk(self.L.apus_gpu_rep_start(self.h, idle_ms, peer_ms, n_append, n_fwork), "rep_start")

    def rep_park(self) -> int: return int(self.L.apus_gpu_rep_park(self.h))

    def rep_submit(self, reqs: np.ndarray, arena: np.ndarray):
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        self._chk(self.L.apus_gpu_rep_submit(self.h, reqs.ctypes.data, len(reqs), arena.ctypes.data, len(arena)), "rep_submit")

    def rep_run(self, r0: int, n: int): self._chk(self.L.apus_gpu_rep_run(self.h, r0, n), "rep_run")

    def rep_prune(self): self._chk(self.L.apus_gpu_rep_prune(self.h), "rep_prune")

    def rep_cmds(self, cmds, repeat: int = 1):
        """a step's commands -- ("run", first staged round, rounds) / ("prune",) -- pushed by ONE call, `repeat` times over"""
        flat = []
        for c in cmds:
            flat += [2, c[1], c[2]] if c[0] == "run" else [1, 0, 0]
        arr = (C.c_uint64 * len(flat))(*flat)
        self._chk(self.L.apus_gpu_rep_cmds(self.h, arr, len(cmds), repeat), "rep_cmds")

    def rep_drain(self, timeout_ms: int = 10000):
        rc = self.L.apus_gpu_rep_drain(self.h, timeout_ms)
        if rc != 0:
            raise EngineError(f"replica kernels did not drain rc={rc} stats={self.rep_stats()} status={self.status_names()}")

    def rep_highest_rec(self) -> int: return int(self.L.apus_gpu_rep_highest_rec(self.h))

    def rep_stats(self) -> dict:
        out = (C.c_uint64 * 8)()
        self.L.apus_gpu_rep_stats(self.h, out)
        d = dict(zip(("rounds", "slots_done", "cmd_head", "commit_slot", "highest_rec", "refused", "dropped", "alive"), [int(v) for v in out]))
        d["exit_code"], d["refused"] = d["refused"] >> 32, d["refused"] & 0xFFFFFFFF      # (exit code of a run that has left: 0 stop, 1 idle, 2 timeout)
        return d

    def rep_latency_ns(self) -> np.ndarray:
        out = np.zeros(1 << 16, dtype=np.uint32)
        n = C.c_uint32(0)
        self._chk(self.L.apus_gpu_rep_latency(self.h, out.ctypes.data, len(out), C.byref(n)), "rep_latency")
        return out[:n.value].copy()

    def rep_latency_appended_ns(self) -> np.ndarray:
        """the round's bytes in every pushed ring -> committed and applied by the leader (SURVEY 8d's round latency)"""
        out = np.zeros(1 << 16, dtype=np.uint32)
        n = C.c_uint32(0)
        self._chk(self.L.apus_gpu_rep_latency_appended(self.h, out.ctypes.data, len(out), C.byref(n)), "rep_latency_appended")
        return out[:n.value].copy()

    def rep_feed(self, reqs: np.ndarray, arena: np.ndarray, n_threads: int, seconds: float, prune_every_reqs: int = 0):
        """n_threads producers on the pinned multi-producer ring for `seconds`, then drained -> (requests, seconds)"""
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        out = (C.c_uint64 * 2)()
        self._chk(self.L.apus_gpu_rep_feed(self.h, reqs.ctypes.data, len(reqs), arena.ctypes.data, len(arena), n_threads, seconds,
                                           prune_every_reqs, out), "rep_feed")
        return int(out[0]), out[1] / 1e9

    def rep_req_ring_kind(self) -> str:
        k = int(self.L.apus_gpu_rep_req_ring_kind(self.h))
        return {1: "device memory, written by the host through the BAR", 0: "pinned host memory, read by the kernel over PCIe"}.get(k, "none yet")

    def rep_launch_ms(self) -> float:
        """duration of the last (parked) run's resident k_replica launch, HIP events on its stream"""
        ms = C.c_double(0)
        self._chk(self.L.apus_gpu_rep_launch_ms(self.h, C.byref(ms)), "rep_launch_ms")
        return ms.value

    def rep_role_stats(self) -> dict:
        """diagnostics of the last run: per serial role passes, passes that moved something, rounds, microseconds"""
        out = np.zeros((20, 8), dtype=np.uint64)
        self._chk(self.L.apus_gpu_rep_role_stats(self.h, out.ctypes.data), "rep_role_stats")
        names = ["sequencer", "committer", "applier"] + [f"f{i}_{w}" for i in range(6) for w in ("retire", "apply")]
        d = {nm: {"passes": int(r[0]), "moved": int(r[1]), "rounds": int(r[2]), "us": int(r[3]) / 100.0, "x": int(r[4]), "busy_us": int(r[5]) / 100.0, "y_us": int(r[6]) / 100.0}
             for nm, r in zip(names, out) if r[0]}
        if "sequencer" in d:
            d["sequencer"].update({"outer_passes": int(out[0][0]), "flow_us": int(out[17][0]) / 100.0, "pcie_us": int(out[17][1]) / 100.0,
                                   "pcie_polls": int(out[17][2]), "reloads": int(out[17][3])})
        if "sequencer" in d and out[18][0]:
            n_p, n_t = max(1, int(out[18][0])), max(1, int(out[18][4]))
            d["sequencer"]["pass_phases_us"] = {"flow": int(out[18][1]) / 100.0 / n_p, "loads": int(out[18][2]) / 100.0 / n_p, "record_and_words": int(out[18][3]) / 100.0 / n_p}
            d["sequencer"]["prune_phases_us"] = {"verify": int(out[18][5]) / 100.0 / n_t, "head_round": int(out[18][6]) / 100.0 / n_t, "rest": int(out[18][7]) / 100.0 / n_t}
        a = out[15]
        if a[0]:
            d["append"] = {"rounds": int(a[0]), "us_per_round": int(a[1]) / 100.0 / int(a[0]), "drain_us": int(a[2]) / 100.0 / int(a[0]),
                           "desc_us": int(a[3]) / 100.0 / int(a[0]), "payload_stores_us": int(a[4]) / 100.0 / int(a[0]),
                           "pre_loop_us": int(a[5]) / 100.0 / int(a[0]), "first_iter_us": int(a[6]) / 100.0 / int(a[0]),
                           "ticket_wait_us": int(a[7]) / 100.0 / int(a[0]), "payload_load_wait_us": int(out[0][7]) / 100.0 / int(a[0]),
                           "load_issue_us": int(out[17][4]) / 100.0 / int(a[0]), "select_us": int(out[17][5]) / 100.0 / int(a[0])}
        w = out[16]
        if w[0]:
            d["f0_work"] = {"rounds": int(w[0]), "us_per_round": int(w[1]) / 100.0 / int(w[0]), "bell_to_headers_us": int(w[2]) / 100.0 / int(w[0]),
                            "stores_issue_us": int(w[3]) / 100.0 / int(w[0]), "drain_us": int(w[4]) / 100.0 / int(w[0])}
        return d

    def rep_roundtrip_ns(self, reqs: np.ndarray, arena: np.ndarray, iters: int) -> np.ndarray:
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        out = np.zeros(iters, dtype=np.uint32)
        self._chk(self.L.apus_gpu_rep_roundtrip(self.h, reqs.ctypes.data, len(reqs), arena.ctypes.data, len(arena), iters, out.ctypes.data),
                  "rep_roundtrip")
        return out

    def run_trace_rep(self, trace: Trace, source: str = "pinned", idle_ms: int = 3000, peer_ms: int = 500,
                      n_append: int = 0, n_fwork: int = 0, drain_each: bool = False, on_event=None):
        """The trace with its ROUND / PRUNE events going through the replica kernels: the leader's and every
        follower's own workgroups.  source = "pinned": the requests cross the pinned multi-producer ring one
        ROUND event at a time (drained per event when drain_each, so that round boundaries are the trace's);
        "staged": stretches of ROUND events run from the staged, device-resident input.  Control-plane events
        (ELECT, HOLD, RELEASE, KILL, JOIN, QUIESCE) park the run and use the control-plane kernels.
        on_event(i, event, engine) is called behind every control-plane event (the run is parked then)."""
        reqs = np.ascontiguousarray(trace.reqs, dtype=REQ_DTYPE)
        arena = np.ascontiguousarray(trace.arena, dtype=np.uint8)
        if source == "staged":
            self.stage_trace(trace)
        running = False

        def park():
            nonlocal running
            if running:
                self.rep_drain()
                code = self.rep_park()
                running = False
                if code != 0:
                    raise EngineError(f"replica kernels exited with code {code}, status {self.status_names()}")
        ev, i = trace.events, 0
        try:
            while i < len(ev):
                op = ev[i][0]
                if op in ("ROUND", "PRUNE"):
                    if not running:
                        self.rep_start(idle_ms, peer_ms, n_append, n_fwork)
                        running = True
                    if op == "PRUNE":
                        self.rep_prune()
                    elif source == "staged":
                        j = i
                        while j < len(ev) and ev[j][0] == "ROUND":
                            j += 1
                        self.rep_run(self.round_of_g0[ev[i][1]], j - i)
                        i = j
                        continue
                    else:
                        self.rep_submit(reqs[ev[i][1]:ev[i][1] + ev[i][2]], arena)
                        if drain_each:
                            self.rep_drain()
                    i += 1
                    continue
                park()
                if op == "ELECT":
                    self.elect(ev[i][1])
                elif op == "QUIESCE":
                    self.quiesce()
                elif op == "HOLD":
                    self.hold(ev[i][1])
                elif op == "RELEASE":
                    self.release(ev[i][1])
                elif op == "KILL":
                    self.kill(ev[i][1])
                elif op == "JOIN":
                    self.join(ev[i][1])
                else:
                    raise EngineError(f"trace event {ev[i]} is not supported")
                if on_event is not None:
                    on_event(i, ev[i], self)
                i += 1
            park()
        finally:
            if running:
                try:
                    self.rep_park()
                except Exception:
                    pass
        self.check_status()

    # -- graphs -------------------------------------------------------------------
    def capture_begin(self): self._chk(self.L.apus_gpu_capture_begin(self.h), "capture_begin")

    def capture_end(self) -> int:
        gid = C.c_int(-1)
        self._chk(self.L.apus_gpu_capture_end(self.h, C.byref(gid)), "capture_end")
        return gid.value

    def graph_launch(self, gid: int): self._chk(self.L.apus_gpu_graph_launch(self.h, gid), "graph_launch")

    # -- observation --------------------------------------------------------------
    def offsets(self, r: int) -> dict:
        out = (C.c_uint64 * 8)()
        self._chk(self.L.apus_gpu_offsets(self.h, r, out), "offsets")
        return dict(zip(OFFSET_NAMES, [int(v) for v in out]))

    def counters(self, r: int) -> dict:
        out = (C.c_uint64 * 8)()
        self._chk(self.L.apus_gpu_counters(self.h, r, out), "counters")
        return dict(zip(COUNTER_NAMES, [int(v) for v in out]))

    def hdr_words(self, r: int) -> np.ndarray:
        out = (C.c_uint64 * 64)()
        self._chk(self.L.apus_gpu_hdr_words(self.h, r, out, 64), "hdr_words")
        return np.array(out[:], dtype=np.uint64)

    def ring(self, r: int, off: int = 0, n: int | None = None) -> np.ndarray:
        n = self.log_len - off if n is None else n
        out = np.empty(n, dtype=np.uint8)
        self._chk(self.L.apus_gpu_read_ring(self.h, r, off, n, out.ctypes.data), "read_ring")
        return out

    def round_record(self):
        n = int(self.L.apus_gpu_round_count(self.h))
        end = np.zeros(n, dtype=np.uint64)
        com = np.zeros(n, dtype=np.uint64)
        if n:
            self._chk(self.L.apus_gpu_round_record(self.h, 0, n, end.ctypes.data, com.ctypes.data), "round_record")
        return com, end

    def apply_records(self, r: int, first: int, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=APPLY_DTYPE)
        if n:
            self._chk(self.L.apus_gpu_apply_records(self.h, r, first, n, out.ctypes.data), "apply_records")
        return out

    def status(self) -> int: return int(self.L.apus_gpu_status(self.h))

    def status_words(self):
        out = (C.c_uint32 * 8)()
        self.L.apus_gpu_status_words.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        self.L.apus_gpu_status_words(self.h, out)
        return [int(v) for v in out]

    def status_names(self):
        try:
            s = self.status()
        except Exception:
            return "?"
        return [v for k, v in ST_NAMES.items() if s & k] or "OK"

    def check_status(self):
        s = self.status()
        if s:
            raise EngineError(f"device status {self.status_names()} words={self.status_words()}")

    def set_timing(self, on: bool): self._chk(self.L.apus_gpu_set_timing(self.h, int(on)), "set_timing")

    def kernel_time(self, which: int = 0):
        ms, n = C.c_float(0), C.c_uint64(0)
        self._chk(self.L.apus_gpu_kernel_time(self.h, which, C.byref(ms), C.byref(n)), "kernel_time")
        return float(ms.value), int(n.value)

    def device_ptr(self, r: int, which: int):
        nb = C.c_uint64(0)
        p = self.L.apus_gpu_device_ptr(self.h, r, which, C.byref(nb))
        return p, int(nb.value)
