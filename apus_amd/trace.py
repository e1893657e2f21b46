"""Event traces for the consensus hot path.

A trace is the totally ordered input shared by the GPU engine, the CPU oracle
and the CPU baseline: admitted client requests (what the proxy's
leader_handle_submit_req produces, /root/reference/src/proxy/proxy.c:108-161)
grouped into leader poll rounds, interleaved with control events.

    ELECT(w)      election won by replica w (also the start-up election)
    ROUND(g0, n)  one leader polling() pass that finds requests [g0, g0+n) queued
    PRUNE         log_pruning timer tick (preceded by an implicit quiesce)
    KILL(r) / HOLD(r) / RELEASE(r) / QUIESCE

Payload bytes of request g are the first `len` bytes of the SplitMix64 stream
seeded with BASE_SEED ^ g (little-endian words).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

BASE_SEED = 0xA9050001
NOOP, CSM, CONFIG, HEAD, CONNECT, SEND, CLOSE = range(7)
HDR = 64
DEFAULT_LOG = 16384 * 4096           # LOG_SIZE, dare_log.h:76
ARENA_FRONT_PAD = 16                 # the device copy reads 2 bytes before a payload
MAX_ROUND = 64                       # one wavefront of entries per device round

REQ_DTYPE = np.dtype([("req_id", "<u8"), ("payload_off", "<u8"), ("clt_id", "<u2"),
                      ("len", "<u2"), ("type", "u1"), ("pad", "u1", (3,))])

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix_words(seeds: np.ndarray, n_words: int) -> np.ndarray:
    """[len(seeds), n_words] uint64: the first n_words outputs of SplitMix64 per seed."""
    seeds = np.asarray(seeds, dtype=np.uint64)
    with np.errstate(over="ignore"):
        k = (np.arange(1, n_words + 1, dtype=np.uint64) * _G)[None, :]
        z = seeds[:, None] + k
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def payload_bytes(seed: int, n: int) -> np.ndarray:
    w = splitmix_words(np.array([seed], dtype=np.uint64), (n + 7) // 8)
    return w.view(np.uint8).reshape(-1)[:n].copy()


@dataclass
class Trace:
    group_size: int
    log_len: int
    reqs: np.ndarray                      # REQ_DTYPE, admission order
    arena: np.ndarray                     # uint8 payload arena (16-B aligned payloads)
    events: list = field(default_factory=list)
    name: str = ""
    leader: int = 0

    @property
    def n_reqs(self) -> int:
        return len(self.reqs)

    def rounds(self):
        return [(a, b) for op, *rest in self.events if op == "ROUND" for a, b in [rest]]

    def entry_bytes(self) -> int:
        """Total log bytes of the client entries (64 + len each)."""
        return int(self.reqs["len"].astype(np.int64).sum() + HDR * len(self.reqs))


class Admission:
    """Mirror of leader_handle_submit_req's id assignment (proxy.c:114-145):
    connection_id = node_id << 8 | pair_count++ (u8 wrap), req_id = per-connection
    counter that starts at 1 with the CONNECT."""

    def __init__(self, node_id: int):
        self.node_id = node_id
        self.pair_count = 0
        self.conns = {}

    def connect(self, fd: int):
        cid = ((self.node_id & 0xFF) << 8) | (self.pair_count & 0xFF)
        self.pair_count = (self.pair_count + 1) & 0xFF
        self.conns[fd] = [cid, 1]
        return cid, 1

    def send(self, fd: int):
        c = self.conns[fd]
        c[1] += 1
        return c[0], c[1]

    def close(self, fd: int):
        c = self.conns.pop(fd)
        return c[0], c[1] + 1


def build_requests(types, fds, lens, leader: int, seed_base: int = BASE_SEED,
                   admission: Admission | None = None, g_base: int = 0):
    """Admit a request stream and materialise its payload arena."""
    n = len(types)
    adm = admission or Admission(leader)
    reqs = np.zeros(n, dtype=REQ_DTYPE)
    lens = np.asarray(lens, dtype=np.int64)
    starts = np.zeros(n, dtype=np.int64)
    pos = ARENA_FRONT_PAD
    # arena layout: every payload 16-B aligned, 16 B slack after the last one
    padded = (lens + 15) // 16 * 16
    starts = ARENA_FRONT_PAD + np.concatenate([[0], np.cumsum(padded)[:-1]]) if n else starts
    total = int(ARENA_FRONT_PAD + padded.sum() + 32)
    arena = np.zeros(total, dtype=np.uint8)
    for g in range(n):
        t = int(types[g])
        if t == CONNECT:
            cid, rid = adm.connect(int(fds[g]))
        elif t == SEND:
            cid, rid = adm.send(int(fds[g]))
        else:
            cid, rid = adm.close(int(fds[g]))
        reqs["req_id"][g] = rid
        reqs["clt_id"][g] = cid
    reqs["type"] = np.asarray(types, dtype=np.uint8)
    reqs["len"] = lens.astype(np.uint16)
    reqs["payload_off"] = starts.astype(np.uint64)
    # payload bytes, vectorised per distinct length
    gidx = np.arange(n, dtype=np.uint64) + np.uint64(g_base)
    seeds = np.uint64(seed_base) ^ gidx
    for L in np.unique(lens):
        L = int(L)
        if L == 0:
            continue
        sel = np.nonzero(lens == L)[0]
        for lo in range(0, len(sel), 1 << 16):
            part = sel[lo:lo + (1 << 16)]
            words = splitmix_words(seeds[part], (L + 7) // 8)
            data = words.view(np.uint8).reshape(len(part), -1)[:, :L]
            idx = starts[part][:, None] + np.arange(L)[None, :]
            arena[idx] = data
    del pos
    return reqs, arena, adm


def _round_events(g0: int, n: int, batch_sizes):
    ev = []
    g = g0
    it = iter(batch_sizes)
    while g < g0 + n:
        b = min(next(it), g0 + n - g)
        ev.append(("ROUND", g, b))
        g += b
    return ev


def steady_trace(group_size: int, n_send: int, payload, conns: int, batch,
                 log_len: int = DEFAULT_LOG, prune_bytes: int | None = None,
                 extra_connects: int = 0, leader: int = 0, name: str = "",
                 seed: int = 0xA9050004) -> Trace:
    """CONNECT x conns, then n_send SEND entries in rounds, with prune ticks.

    payload: int (fixed) or sequence of choices (drawn by SplitMix64(seed));
    batch: int (fixed) or (lo, hi) drawn from the same stream."""
    n_conn = conns + extra_connects
    rng_state = np.array([seed], dtype=np.uint64)

    def draws(k):
        nonlocal rng_state
        w = splitmix_words(rng_state, k)[0]
        with np.errstate(over="ignore"):
            rng_state = rng_state + np.uint64(k) * _G
        return w

    if isinstance(payload, (int, np.integer)):
        lens_send = np.full(n_send, int(payload), dtype=np.int64)
    else:
        choices = np.asarray(list(payload), dtype=np.int64)
        lens_send = choices[(draws(n_send) % np.uint64(len(choices))).astype(np.int64)]
    types = np.concatenate([np.full(n_conn, CONNECT), np.full(n_send, SEND)]).astype(np.uint8)
    fds = np.concatenate([np.arange(n_conn), np.arange(n_send) % n_conn]).astype(np.int64) + 100
    lens = np.concatenate([np.zeros(n_conn, dtype=np.int64), lens_send])
    reqs, arena, _ = build_requests(types, fds, lens, leader)

    if isinstance(batch, (int, np.integer)):
        def batch_iter():
            while True:
                yield int(batch)
    else:
        lo, hi = batch
        def batch_iter():
            while True:
                for v in draws(1024):
                    yield int(lo + int(v) % (hi - lo + 1))

    if prune_bytes is None:
        prune_bytes = 8 << 20 if log_len >= (64 << 20) else max(log_len // 8, 1024)
    events = [("ELECT", leader)]
    events += _round_events(0, n_conn, batch_iter())
    since = HDR * (n_conn + 1)
    g = n_conn
    bi = batch_iter()
    n = len(reqs)
    elen = lens + HDR
    while g < n:
        b = min(next(bi), n - g, MAX_ROUND)
        events.append(("ROUND", g, b))
        since += int(elen[g:g + b].sum())
        g += b
        if since >= prune_bytes:
            events.append(("PRUNE",))
            since = 0
    events.append(("QUIESCE",))
    return Trace(group_size, log_len, reqs, arena, events, name or f"steady_n{group_size}", leader)


# the configurations of BASELINE.json, full size unless scaled down by the caller
def config_c2(n_send: int = 1 << 20, group_size: int = 3, log_len: int = DEFAULT_LOG, **kw) -> Trace:
    return steady_trace(group_size, n_send, 64, 16, 64, log_len, name="C2", **kw)


def config_c3(n_send: int = 1 << 18, group_size: int = 5, log_len: int = DEFAULT_LOG, **kw) -> Trace:
    return steady_trace(group_size, n_send, 1024, 16, 32, log_len, name="C3", **kw)


def config_c4(n_send: int = 1 << 18, group_size: int = 7, log_len: int = DEFAULT_LOG, **kw) -> Trace:
    return steady_trace(group_size, n_send, (64, 128, 256, 512, 1024, 2048, 4096), 64, (1, 64),
                        log_len, name="C4", **kw)


def config_c5(per_phase: int = 20000, group_size: int = 5, log_len: int = DEFAULT_LOG,
              conns: int = 16, batch: int = 16, rejoin: bool = False, prune_bytes: int | None = None) -> Trace:
    """reconf_bench.sh shape (benchmarks/reconf_bench.sh:249-343): bench, kill the
    leader, bench, kill one follower, bench.  SET/GET-sized 107/40-byte requests.
    rejoin: the optional tail -- a new server joins into the lowest empty slot (the dead leader's), recovers
    the log from the group and a fourth phase runs on four servers again."""
    phases = 4 if rejoin else 3
    if prune_bytes is None:
        prune_bytes = max(log_len // 8, 1024)
    n_send = per_phase * phases
    lens_send = np.where(np.arange(n_send) % 2 == 0, 107, 40).astype(np.int64)
    # connections are re-established against the new leader after the fail-over
    events = []
    all_reqs, arenas = [], []
    arena_off = 0
    g = 0
    leader = 0
    alive = list(range(group_size))
    for ph in range(phases):
        if ph == 0:
            events.append(("ELECT", leader))
        elif ph == 1:
            events.append(("KILL", leader))
            alive.remove(leader)
            leader = alive[0]
            events.append(("ELECT", leader))
        elif ph == 2:
            victim = alive[-1]
            events.append(("KILL", victim))
            alive.remove(victim)
        else:
            events += [("JOIN", 0), ("QUIESCE",)]
        lens_p = lens_send[ph * per_phase:(ph + 1) * per_phase]
        if ph < 2:
            types = np.concatenate([np.full(conns, CONNECT), np.full(per_phase, SEND)]).astype(np.uint8)
            fds = np.concatenate([np.arange(conns), np.arange(per_phase) % conns]) + 100
            lens = np.concatenate([np.zeros(conns, dtype=np.int64), lens_p])
            adm = Admission(leader)
        else:
            types = np.full(per_phase, SEND, dtype=np.uint8)
            fds = np.arange(per_phase) % conns + 100
            lens = lens_p
        reqs, arena, adm = build_requests(types, fds, lens, leader, admission=adm, g_base=g)
        reqs["payload_off"] += np.uint64(arena_off)
        all_reqs.append(reqs)
        arenas.append(arena)
        arena_off += len(arena)
        n = len(reqs)
        k = 0
        since = 0
        while k < n:
            b = min(batch, n - k)
            events.append(("ROUND", g + k, b))
            since += int((lens[k:k + b] + HDR).sum())
            k += b
            if since >= prune_bytes:
                events.append(("PRUNE",))
                since = 0
        g += n
        events.append(("QUIESCE",))
    return Trace(group_size, log_len, np.concatenate(all_reqs), np.concatenate(arenas), events, "C5", 0)
