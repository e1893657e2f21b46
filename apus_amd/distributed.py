"""One replica per GPU / process: the quorum round over RCCL point-to-point.

Replaces the reference's RDMA data plane (src/dare/dare_ibv_rc.c) between
processes.  Rank r hosts replica r on its own GPU; rank `leader` runs the leader
half of the engine, every other rank a follower half.  Per leader batch:

    R1  WRITE log bytes  [follower end, leader end)  -> send of the ring range
        (a wrapped range is two pieces, like the two WRs of dare_ibv_rc.c:1538-1545)
        plus the matching directory slots (derived data, 12 B per entry)
    R2  WRITE end                                     -> the header of that message
    R3  1-byte ACK per entry (rc_send_entries_reply)  -> ONE cumulative slot number
        per follower and batch; the leader expands it into reply bytes + ACK bits
        (apus_gpu_ack_merge) so that the ACK scan reads the same words as always
    R4  lazy WRITE commit (dare_ibv_rc.c:1761-1819)   -> piggy-backed on the next header
    R8  READ apply offset (rc_get_remote_apply_offsets) -> rides on the ACK reply

`backend="nccl"` is RCCL: device tensors that alias the HBM rings are sent
directly (xGMI peer-to-peer, no staging).  `backend="gloo"` stages through host
memory and exists for the CPU/one-GPU tests.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .engine import Engine, EngineError
from .trace import DEFAULT_LOG

OP_DATA, OP_STOP, OP_MARK = 1, 2, 3
HDR_WORDS, REPLY_WORDS = 16, 4
H_APPLY_OFFSETS = 23          # apus_device.h


class _DevMem:
    """Zero-copy torch view of engine-owned device memory."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_view(eng: Engine, replica: int, which: int, device) -> torch.Tensor:
    ptr, nb = eng.device_ptr(replica, which)
    return torch.as_tensor(_DevMem(ptr, nb), device=device)


def ring_pieces(frm: int, to: int, length: int):
    """Byte pieces of the circular range [frm, to) of a ring of `length` bytes."""
    if frm == length:
        frm = 0
    if to == frm:
        return []
    if to > frm:
        return [(frm, to)]
    return [(frm, length), (0, to)] if to > 0 else [(frm, length)]


def slot_pieces(s0: int, s1: int, cap: int):
    """Index pieces of directory slots [s0, s1) in a directory ring of `cap` slots."""
    out = []
    while s0 < s1:
        i = s0 % cap
        n = min(s1 - s0, cap - i)
        out.append((i, i + n))
        s0 += n
    return out


class Transport:
    """send/recv of byte tensors that may live on the GPU."""

    def __init__(self, backend: str, device):
        self.backend = backend
        self.device = device
        self.direct = backend == "nccl"

    def _meta(self, words):
        t = torch.tensor(words, dtype=torch.int64)
        return t.to(self.device) if self.direct else t

    def send_words(self, words, dst):
        dist.send(self._meta(words), dst)

    def recv_words(self, n, src):
        t = torch.zeros(n, dtype=torch.int64, device=self.device if self.direct else "cpu")
        dist.recv(t, src)
        return [int(v) for v in t.cpu().tolist()]

    def send_bytes(self, t: torch.Tensor, dst):
        dist.send(t if self.direct else t.cpu(), dst)

    def recv_bytes(self, view: torch.Tensor, src):
        if self.direct or view.device.type == "cpu":
            dist.recv(view, src)
        else:
            tmp = torch.empty(view.shape, dtype=view.dtype)
            dist.recv(tmp, src)
            view.copy_(tmp)

    # ---- batched point-to-point (one ncclGroup per call on RCCL: all links at once) ----
    def batch_send(self, items):
        """items: [(tensor, dst)]"""
        if not items:
            return
        keep = [(t if self.direct else t.cpu().contiguous(), d) for t, d in items]
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, d) for t, d in keep]):
            w.wait()

    def batch_recv(self, items):
        """items: [(view, src)]; data lands in the views"""
        if not items:
            return
        if self.direct:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, v, s_) for v, s_ in items]):
                w.wait()
            return
        tmps = [(torch.empty(v.shape, dtype=v.dtype), v, s_) for v, s_ in items]
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, t, s_) for t, _, s_ in tmps]):
            w.wait()
        for t, v, _ in tmps:
            v.copy_(t)

    def fence(self):
        """Received bytes are in HBM before the engine's own stream touches them: RCCL
        completes on torch's stream, the engine launches on its own non-blocking one."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)


def range_items(ring, dir_off, dir_len, o0: int, o1: int, s0: int, s1: int, log_len: int, dir_cap: int):
    """The tensors that make up R1 for slots [s0, s1): ring pieces, then directory pieces."""
    if s1 <= s0:
        return []
    out = [ring[a:b] for a, b in ring_pieces(o0, o1, log_len)]
    for a, b in slot_pieces(s0, s1, dir_cap):
        out.append(dir_off[8 * a:8 * b])
        out.append(dir_len[4 * a:4 * b])
    return out


def header_words(o0, o1, s0, s1, commit_slot, term):
    return [OP_DATA, o0, o1, s0, max(s1, s0), commit_slot, term, 0] + [0] * 8


def ship_range(tp: Transport, dst: int, ring, dir_off, dir_len, o0: int, o1: int, s0: int, s1: int,
               log_len: int, dir_cap: int, commit_slot: int, term: int):
    """Leader side of R1+R2 for one follower: header, ring pieces, directory pieces.
    ring / dir_off / dir_len are byte tensors (device or host)."""
    tp.send_words(header_words(o0, o1, s0, s1, commit_slot, term), dst)
    tp.batch_send([(t, dst) for t in range_items(ring, dir_off, dir_len, o0, o1, s0, s1, log_len, dir_cap)])


def recv_range(tp: Transport, src: int, ring, dir_off, dir_len, o0: int, o1: int, s0: int, s1: int,
               log_len: int, dir_cap: int):
    """Follower side: the bytes land at the same offsets of the local ring / directory."""
    tp.batch_recv([(t, src) for t in range_items(ring, dir_off, dir_len, o0, o1, s0, s1, log_len, dir_cap)])
    tp.fence()


class GroupMember:
    """The replica hosted by this rank, plus its share of the exchange."""

    def __init__(self, group_size: int, rank: int, leader: int, device_index: int, backend: str,
                 log_len: int = DEFAULT_LOG):
        self.n, self.rank, self.leader = group_size, rank, leader
        self.device = torch.device("cuda", device_index)
        self.eng = Engine(group_size, log_len, local_ids=[rank], device=device_index)
        self.log_len = log_len
        self.tp = Transport(backend, self.device)
        self.ring = device_view(self.eng, rank, 0, self.device)
        self.dir_off = device_view(self.eng, rank, 2, self.device)
        self.dir_len = device_view(self.eng, rank, 3, self.device)
        self.dir_cap = self.dir_len.numel() // 4
        self.is_leader = rank == leader
        self.followers = [r for r in range(group_size) if r != leader]
        # leader-side view of every follower
        self.shipped_slot = {f: 0 for f in self.followers}
        self.shipped_off = {f: log_len for f in self.followers}     # log_len == empty
        self.acked = {f: 0 for f in self.followers}
        self.term = 0
        self.marks = []

    # ---------------------------------------------------------------- leader
    def elect(self):
        self.term += 2
        mask = (1 << self.n) - 1
        if self.is_leader:
            self.eng.elect(self.leader)            # sets term, blank CONFIG entry
        else:
            self.eng.term = self.term
            self.eng._chk(self.eng.L.apus_gpu_follow(self.eng.h, self.rank, self.leader, self.term, mask), "follow")

    def _leader_state(self):
        """(visible slot, its byte offset, commit slot) of the leader: one read-back."""
        w = self.eng.hdr_words(self.rank)
        end, n_end, n_commit, n_vis = int(w[3]), int(w[8]), int(w[10]), int(w[22])
        vis = n_vis if end == self.log_len else n_end          # end == len: the log reads as empty
        if vis == n_end:
            vis_off = end
        else:
            vis_off = int(self.dir_off.view(torch.int64)[vis % self.dir_cap].item())
        return vis, vis_off, n_commit

    def sync_followers(self):
        """Ship what the followers lack (R1+R2), collect the ACKs (R3), merge them.
        Every follower is served in the same batch (one link each on xGMI)."""
        vis, vis_off, n_commit = self._leader_state()
        hdrs, data = [], []
        for f in self.followers:
            s0, o0 = self.shipped_slot[f], self.shipped_off[f]
            hdrs.append((self.tp._meta(header_words(o0, vis_off, s0, vis, n_commit, self.term)), f))
            data += [(t, f) for t in range_items(self.ring, self.dir_off, self.dir_len, o0, vis_off, s0, vis,
                                                 self.log_len, self.dir_cap)]
            if vis > s0:
                self.shipped_slot[f], self.shipped_off[f] = vis, vis_off
        self.tp.batch_send(hdrs)
        self.tp.batch_send(data)
        rep = {f: torch.zeros(REPLY_WORDS, dtype=torch.int64, device=self.device if self.tp.direct else "cpu")
               for f in self.followers}
        self.tp.batch_recv([(rep[f], f) for f in self.followers])
        replies = {}
        for f in self.followers:
            r = [int(v) for v in rep[f].cpu().tolist()]
            replies[f] = r
            if r[3]:
                raise EngineError(f"follower {f} reports device status {r[3]:#x}")
            if r[0] > self.acked[f]:
                self.eng._chk(self.eng.L.apus_gpu_ack_merge(self.eng.h, f, self.acked[f], r[0]), "ack_merge")
                self.acked[f] = r[0]
        return replies

    def leader_rounds(self, r0: int, n_rounds: int):
        self.eng._chk(self.eng.L.apus_gpu_append_rounds(self.eng.h, r0, n_rounds), "append_rounds")
        self.sync_followers()
        self.eng._chk(self.eng.L.apus_gpu_commit_rounds(self.eng.h, r0, n_rounds), "commit_rounds")

    def leader_quiesce(self):
        """Followers learn the newest commit and apply it (the lazy R4 made eager)."""
        self.sync_followers()
        self.eng.quiesce()
        return self.sync_followers()

    def leader_prune(self):
        replies = self.leader_quiesce()
        self.eng.tick_prune()                                  # decision + maybe a HEAD entry
        # R8: the apply offsets just read become the input of the next tick.  The tick runs on the
        # engine's own stream, the stores below on torch's: drain the first before, the second after
        self.eng.sync()
        hdr_ptr, _ = self.eng.device_ptr(self.rank, 1)
        hdr = torch.as_tensor(_DevMem(hdr_ptr, 64 * 8), device=self.device).view(torch.int64)
        for f, r in replies.items():
            hdr[H_APPLY_OFFSETS + f] = r[2]
        torch.cuda.synchronize(self.device)
        self.sync_followers()                                  # ship the HEAD entry, if any
        self.eng.quiesce()

    def leader_stop(self):
        for f in self.followers:
            self.tp.send_words([OP_STOP] + [0] * (HDR_WORDS - 1), f)

    def mark(self):
        """Timing mark on every rank: device sync + barrier, then a timestamp."""
        if self.is_leader:
            for f in self.followers:
                self.tp.send_words([OP_MARK] + [0] * (HDR_WORDS - 1), f)
        self.eng.sync()
        torch.cuda.synchronize(self.device)
        dist.barrier()
        self.marks.append(time.perf_counter())

    # -------------------------------------------------------------- follower
    def follower_serve(self):
        """Serve the leader until it says stop."""
        eng, L = self.eng, self.eng.L
        while True:
            h = self.tp.recv_words(HDR_WORDS, self.leader)
            if h[0] == OP_STOP:
                return
            if h[0] == OP_MARK:
                self.mark()
                continue
            _, o0, o1, s0, s1, commit, term, npieces = h[:8]
            if s1 > s0:
                recv_range(self.tp, self.leader, self.ring, self.dir_off, self.dir_len, o0, o1, s0, s1,
                           self.log_len, self.dir_cap)
                eng._chk(L.apus_gpu_ingest(eng.h, self.rank, s1, s1 - s0), "ingest")
            eng._chk(L.apus_gpu_follower_commit(eng.h, self.rank, commit, max(s1 - s0, 1)), "follower_commit")
            w = eng.hdr_words(self.rank)               # synchronises: the ACK means "persisted"
            self.tp.send_words([int(w[9]), int(w[11]), int(w[1]), eng.status()], self.leader)

    def close(self):
        self.eng.close()


# ------------------------------------------------------------------------------------
def run_trace_group(member: GroupMember, trace):
    """Drive one trace through a multi-process group (steady-state events)."""
    if member.is_leader:
        member.eng.stage_trace(trace)
    member.elect()
    if not member.is_leader:
        member.follower_serve()
        return
    member.sync_followers()            # the blank CONFIG entry
    member.eng.quiesce()
    ev, i = trace.events, 0
    while i < len(ev):
        if ev[i][0] == "ROUND":
            j = i
            while j < len(ev) and ev[j][0] == "ROUND":
                j += 1
            member.leader_rounds(member.eng.round_of_g0[ev[i][1]], j - i)
            i = j
            continue
        if ev[i][0] == "PRUNE":
            member.leader_prune()
        elif ev[i][0] == "QUIESCE":
            member.leader_quiesce()
        elif ev[i][0] != "ELECT":
            raise EngineError(f"{ev[i]} is not supported in multi-process groups yet")
        i += 1
    member.leader_quiesce()
    member.leader_stop()


def bench_group(args, initialised=None, n_rep=None, keep_group=False, entries=None, steps=None, warmup=None):
    """The message-passing twin of the peer-mapped data plane: N replicas, one per GPU, R1 / R2 as RCCL send / recv of the ring
    range a follower lacks, R3 as one cumulative word back (module docstring).  Two uses:
      * bench.py --gpus N with APUS_GROUP_TRANSPORT=p2p, or where the devices cannot map each other's memory: the line;
      * inside bench.py --gpus N's normal run (keep_group=True, a shorter workload): the `rccl_transport` object next to the
        peer-mapped headline -- north_star names both ways of carrying a round between GPUs.
    n_rep < world: the ranks behind n_rep are spare machines; they only stand in the barriers and reductions."""
    from . import trace as T
    if initialised is None:
        from .peers import init_process_group_from_env
        initialised = init_process_group_from_env(args.gpus)
    rank, world, local, backend = initialised
    n = n_rep or world
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    entries = entries or args.entries
    spare = rank >= n
    red_dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    tr = T.steady_trace(n, entries, args.payload, 16, args.batch, log_len=T.DEFAULT_LOG, name="C2")
    n_entries = len(tr.reqs)
    good, dt = True, 0.0
    m = None
    if spare:
        dist.barrier()                      # (the two marks of the timed region)
        dist.barrier()
    else:
        m = GroupMember(n, rank, 0, local, backend, tr.log_len)
        calls = None
        if m.is_leader:
            m.eng.stage_trace(tr)
            ev, i, calls = tr.events, 0, []
            while i < len(ev):
                if ev[i][0] == "ROUND":
                    j = i
                    while j < len(ev) and ev[j][0] == "ROUND":
                        j += 1
                    calls.append(("rounds", m.eng.round_of_g0[ev[i][1]], j - i))
                    i = j
                    continue
                if ev[i][0] == "PRUNE":
                    calls.append(("prune",))
                i += 1
        m.elect()

        def leader_step():
            for c in calls:
                if c[0] == "rounds":
                    m.leader_rounds(c[1], c[2])
                else:
                    m.leader_prune()
            m.leader_quiesce()

        if m.is_leader:
            m.sync_followers()
            m.eng.quiesce()
            for _ in range(warmup):
                leader_step()
            # exactly K steps between two marks; a mark = device sync + barrier on every rank
            m.mark()
            for _ in range(steps):
                leader_step()
            m.mark()
            m.leader_stop()
        else:
            m.follower_serve()
        dt = m.marks[1] - m.marks[0]
        o = m.eng.offsets(rank)
        total = (warmup + steps) * n_entries
        applied = m.eng.counters(rank)["highest_rec"] if m.is_leader else int(m.eng.hdr_words(rank)[16])
        good = (o["commit"] == o["end"] == o["apply"]) and applied == total and not m.eng.status()
    # max over ranks
    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ok = torch.tensor([1.0 if good else 0.0], device=red_dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    out = None
    if rank == 0:
        E = 64 + args.payload
        value = n_entries * steps / dt
        out = {
            "metric": "committed entries/sec", "value": value, "unit": "entries/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{n} replicas, one per GPU ({'RCCL p2p over xGMI' if backend == 'nccl' else backend + ' staging (test mode)'}), {n_entries} entries/step "
                                   f"of {args.payload} B, rounds of {args.batch}, prune tick every 8 MiB",
                       "mode": "one process per replica GPU", "replicas": n, "entry_bytes": E},
            "verified": bool(ok.item() == 1),
            "roofline": {"bound": "xgmi", "achieved": value * E / 1e9, "peak": 153.0, "unit": "GB/s",
                         "frac": value * E / 1e9 / 153.0, "traffic": None,
                         "note": "per leader->follower link: entries/s x E against one xGMI link (SURVEY.md 8d)"},
        }
    if m is not None:
        m.close()
    if not keep_group:
        dist.destroy_process_group()
    return out
