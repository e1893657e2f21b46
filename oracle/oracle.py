"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present,
the reference's own dare_log.h compiled unchanged (oracle/_ref/libapus_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the apus_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libapus_ref.so")

MAX_SERVERS = 13
HDR = 64
DEFAULT_LOG = 16384 * 4096
NOOP, CSM, CONFIG, HEAD, CONNECT, SEND, CLOSE = range(7)

u64, u32, u16, u8 = C.c_uint64, C.c_uint32, C.c_uint16, C.c_uint8
vp = C.c_void_p

REQ_DTYPE = np.dtype([("req_id", "<u8"), ("payload_off", "<u8"), ("clt_id", "<u2"),
                      ("len", "<u2"), ("type", "u1"), ("pad", "u1", (3,))])
APPLY_DTYPE = np.dtype([("slot", "<u8"), ("off", "<u8"), ("idx", "<u8"), ("len", "<u4"),
                        ("clt_id", "<u2"), ("type", "u1"), ("kind", "u1")])
assert REQ_DTYPE.itemsize == 24 and APPLY_DTYPE.itemsize == 32


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "apus_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/include/dare") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        sig = {
            "orc_log_new": (vp, [u64]),
            "orc_log_free": (None, [vp]),
            "orc_log_append": (u64, [vp, u64, u64, u16, u8, vp, u16]),
            "orc_log_offsets": (None, [vp, C.POINTER(u64)]),
            "orc_log_set_offsets": (None, [vp, C.POINTER(u64)]),
            "orc_log_entries": (vp, [vp]),
            "orc_log_prev_head": (C.c_int, [vp]),
            "orc_log_set_prev_head": (None, [vp, C.c_int]),
            "orc_log_end_distance": (u64, [vp, u64]),
            "orc_log_is_larger": (C.c_int, [vp, u64, u64]),
            "orc_log_get_entry": (u64, [vp, u64]),
            "orc_log_entry_len_at": (u32, [vp, u64]),
            "orc_log_get_tail": (u64, [vp]),
            "orc_log_to_ncbuf": (None, [vp, vp]),
            "orc_log_find_remote_end": (u64, [vp, vp]),
            "orc_cluster_new": (vp, [C.c_int, u64]),
            "orc_cluster_free": (None, [vp]),
            "orc_cluster_record_apply": (None, [vp, C.c_int]),
            "orc_cluster_allow_exact_fit": (None, [vp, C.c_int]),
            "orc_cluster_completion_delay": (None, [vp, C.c_int]),
            "orc_force_prune_count": (u64, [vp]),
            "orc_elect": (C.c_int, [vp, C.c_int]),
            "orc_round": (C.c_int, [vp, vp, C.c_int, vp]),
            "orc_tick_prune": (C.c_int, [vp]),
            "orc_kill": (C.c_int, [vp, C.c_int]),
            "orc_hold": (C.c_int, [vp, C.c_int]),
            "orc_release": (C.c_int, [vp, C.c_int]),
            "orc_quiesce": (C.c_int, [vp]),
            "orc_join": (C.c_int, [vp, C.c_int]),
            "orc_cluster_record_store": (None, [vp, C.c_int]),
            "orc_replica_store_stream": (vp, [vp, C.c_int, C.POINTER(u64)]),
            "orc_replica_records_len": (u32, [vp, C.c_int]),
            "orc_replica_cid": (None, [vp, C.c_int, vp]),
            "orc_replica_alive": (C.c_int, [vp, C.c_int]),
            "orc_leader": (C.c_int, [vp]),
            "orc_group_size": (C.c_int, [vp]),
            "orc_replica_log": (vp, [vp, C.c_int]),
            "orc_replica_sid": (u64, [vp, C.c_int]),
            "orc_replica_cid_bitmask": (C.c_uint32, [vp, C.c_int]),
            "orc_replica_highest_rec": (u64, [vp, C.c_int]),
            "orc_replica_apply_count": (u64, [vp, C.c_int]),
            "orc_replica_apply_hash": (u64, [vp, C.c_int]),
            "orc_replica_store_count": (u64, [vp, C.c_int]),
            "orc_replica_apply_log": (vp, [vp, C.c_int, C.POINTER(u64)]),
            "orc_round_count": (u64, [vp]),
            "orc_round_commit": (C.POINTER(u64), [vp]),
            "orc_round_end": (C.POINTER(u64), [vp]),
            "orc_run_rounds": (C.c_int, [vp, vp, vp, u64, vp, u64]),
            "orc_run_rounds_mt": (C.c_int, [vp, vp, vp, u64, vp, u64, C.c_double]),
            "orc_fill_payload": (None, [u64, vp, u32]),
            "orc_apply_mix": (u64, [u64, u64, u64, u32, u16, u8, u8]),
            "orc_canon": (u64, [vp, u64, u64, u64, u64, vp, u64, C.POINTER(u64)]),
            "orc_canon_hash": (u64, [vp, u64, u64, u64, u64, C.POINTER(u64)]),
            "orc_defined_mask": (u64, [vp, u64, u64, u64, u64, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def have_ref() -> bool:
    if os.path.isdir("/root/reference/src/include/dare"):
        build()
    return os.path.exists(REF_SO)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libapus_ref.so is not built and /root/reference is absent")
        R = C.CDLL(REF_SO)
        sig = {
            "ref_log_new": (vp, [u64]),
            "ref_log_free": (None, [vp]),
            "ref_log_append": (u64, [vp, u64, u64, u16, u8, vp, u16]),
            "ref_log_offsets": (None, [vp, C.POINTER(u64)]),
            "ref_log_set_offsets": (None, [vp, C.POINTER(u64)]),
            "ref_log_entries": (vp, [vp]),
            "ref_prev_head": (C.c_int, []),
            "ref_set_prev_head": (None, [C.c_int]),
            "ref_log_end_distance": (u64, [vp, u64]),
            "ref_log_is_larger": (C.c_int, [vp, u64, u64]),
            "ref_log_get_entry": (u64, [vp, u64]),
            "ref_log_entry_len_at": (u32, [vp, u64]),
            "ref_log_get_tail": (u64, [vp]),
            "ref_log_to_ncbuf": (u64, [vp, C.c_int, C.POINTER(u64), u64]),
            "ref_log_find_remote_end": (u64, [vp, C.c_int, C.POINTER(u64), u64]),
            "ref_layout": (None, [C.POINTER(u64)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(R, name)
            fn.restype = res
            fn.argtypes = args
        _ref = R
    return _ref


def _data_arg(type_: int, data) -> tuple:
    """Marshal the `data` argument of an append: bytes payload, 16-B cid or u64 head."""
    if data is None:
        return None, 0
    if isinstance(data, int):
        buf = (u64 * 1)(data)
        return C.cast(buf, vp), 0
    b = bytes(data)
    buf = C.create_string_buffer(b, len(b) + 1)
    return C.cast(buf, vp), (0 if type_ in (NOOP, CONFIG, HEAD) else len(b))


class _LogBase:
    """Common Python face of one log (oracle restatement or reference build)."""
    pfx = ""
    L = None

    def __init__(self, handle, owned: bool):
        self.h = handle
        self.owned = owned

    def _f(self, name):
        return getattr(self.L, self.pfx + name)

    def append(self, term, req_id, clt_id, type_, data=None) -> int:
        ptr, n = _data_arg(type_, data)
        return self._f("log_append")(self.h, term, req_id, clt_id, type_, ptr, n)

    def offsets(self) -> dict:
        out = (u64 * 8)()
        self._f("log_offsets")(self.h, out)
        return dict(zip(("head", "apply", "commit", "end", "tail", "old_end", "old_commit", "len"),
                        [int(v) for v in out]))

    def set_offsets(self, **kw) -> None:
        o = self.offsets()
        o.update(kw)
        arr = (u64 * 8)(*[o[k] for k in ("head", "apply", "commit", "end", "tail", "old_end",
                                           "old_commit", "len")])
        self._f("log_set_offsets")(self.h, arr)

    def ring(self) -> np.ndarray:
        """Zero-copy numpy view of entries[0:len]."""
        n = self.offsets()["len"]
        ptr = self._f("log_entries")(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(u8)), shape=(n,))

    def end_distance(self, off): return int(self._f("log_end_distance")(self.h, off))
    def is_larger(self, l, r): return int(self._f("log_is_larger")(self.h, l, r))

    def get_entry(self, off):
        v = int(self._f("log_get_entry")(self.h, off))
        return None if v == 2**64 - 1 else v

    def entry_len_at(self, off): return int(self._f("log_entry_len_at")(self.h, off))
    def get_tail(self): return int(self._f("log_get_tail")(self.h))


class OracleLog(_LogBase):
    pfx = "orc_"

    def __init__(self, length: int = DEFAULT_LOG, handle=None):
        self.L = lib()
        if handle is None:
            super().__init__(self.L.orc_log_new(length), True)
        else:
            super().__init__(handle, False)

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.L.orc_log_free(self.h)
            self.h = None

    @property
    def prev_head(self): return int(self.L.orc_log_prev_head(self.h))

    @prev_head.setter
    def prev_head(self, v): self.L.orc_log_set_prev_head(self.h, int(v))

    def to_ncbuf(self):
        buf = (u64 * (1 + 3 * 1024))()
        self.L.orc_log_to_ncbuf(self.h, buf)
        n = int(buf[0])
        return [(int(buf[1 + 3 * i]), int(buf[2 + 3 * i]), int(buf[3 + 3 * i])) for i in range(n)]

    def find_remote_end(self, dets):
        buf = (u64 * (1 + 3 * 1024))()
        buf[0] = len(dets)
        for i, (a, b, c) in enumerate(dets):
            buf[1 + 3 * i], buf[2 + 3 * i], buf[3 + 3 * i] = a, b, c
        return int(self.L.orc_log_find_remote_end(self.h, buf))


class RefLog(_LogBase):
    """The reference's dare_log.h itself (one live instance at a time: the header
    keeps prev_log_entry_head in a process-wide global)."""
    pfx = "ref_"

    def __init__(self, length: int = DEFAULT_LOG):
        self.L = ref()
        self.L.ref_set_prev_head(0)
        super().__init__(self.L.ref_log_new(length), True)

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.L.ref_log_free(self.h)
            self.h = None

    @property
    def prev_head(self): return int(self.L.ref_prev_head())

    @prev_head.setter
    def prev_head(self, v): self.L.ref_set_prev_head(int(v))

    def to_ncbuf(self, slot: int = 0):
        buf = (u64 * (3 * 1024))()
        n = int(self.L.ref_log_to_ncbuf(self.h, slot, buf, 1024))
        return [(int(buf[3 * i]), int(buf[3 * i + 1]), int(buf[3 * i + 2])) for i in range(n)]

    def find_remote_end(self, dets, slot: int = 1):
        buf = (u64 * max(3 * len(dets), 3))()
        for i, (a, b, c) in enumerate(dets):
            buf[3 * i], buf[3 * i + 1], buf[3 * i + 2] = a, b, c
        return int(self.L.ref_log_find_remote_end(self.h, slot, buf, len(dets)))


def ref_layout() -> dict:
    out = (u64 * 12)()
    ref().ref_layout(out)
    keys = ("sizeof_entry", "idx", "term", "req_id", "clt_id", "type", "sender", "reply", "data",
            "sizeof_cid", "entries", "LOG_SIZE")
    return dict(zip(keys, [int(v) for v in out]))


def cid_bytes(epoch: int, size0: int, size1: int, state: int, bitmask: int) -> bytes:
    """dare_cid_t as 16 little-endian bytes."""
    import struct
    return struct.pack("<QBBBBI", epoch, size0, size1, state, 0, bitmask)


class Cluster:
    """N in-process replicas driven by trace events."""

    def __init__(self, group_size: int, log_len: int = DEFAULT_LOG, record_apply: bool = True,
                 allow_exact_fit: bool = True, completion_delay: bool = False):
        self.L = lib()
        self.h = self.L.orc_cluster_new(group_size, log_len)
        if not self.h:
            raise ValueError("bad group size")
        self.n = group_size
        self.log_len = log_len
        self.L.orc_cluster_record_apply(self.h, int(record_apply))
        self.L.orc_cluster_allow_exact_fit(self.h, int(allow_exact_fit))
        self.L.orc_cluster_completion_delay(self.h, int(completion_delay))
        self.L.orc_cluster_record_store(self.h, int(record_apply))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_cluster_free(self.h)
            self.h = None

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"oracle {what} failed rc={rc}")
        return rc

    def elect(self, winner): return self._chk(self.L.orc_elect(self.h, winner), "elect")
    def tick_prune(self): return self._chk(self.L.orc_tick_prune(self.h), "tick_prune")
    def kill(self, r): return self._chk(self.L.orc_kill(self.h, r), "kill")
    def hold(self, r): return self._chk(self.L.orc_hold(self.h, r), "hold")
    def release(self, r): return self._chk(self.L.orc_release(self.h, r), "release")
    def quiesce(self): return self._chk(self.L.orc_quiesce(self.h), "quiesce")

    def join(self, r):
        rc = self._chk(self.L.orc_join(self.h, r), "join")
        self.n = int(self.L.orc_group_size(self.h))
        return rc

    def alive(self, r): return bool(self.L.orc_replica_alive(self.h, r))

    def record_store(self, on=True): self.L.orc_cluster_record_store(self.h, int(on))

    def store_stream(self, r) -> bytes:
        n = u64(0)
        ptr = self.L.orc_replica_store_stream(self.h, r, C.byref(n))
        return C.string_at(ptr, n.value) if n.value else b""

    def records_len(self, r): return int(self.L.orc_replica_records_len(self.h, r))

    def cid(self, r) -> dict:
        buf = (C.c_uint8 * 16)()
        self.L.orc_replica_cid(self.h, r, buf)
        b = bytes(buf)
        return {"epoch": int.from_bytes(b[0:8], "little"), "size0": b[8], "size1": b[9], "state": b[10],
                "bitmask": int.from_bytes(b[12:16], "little")}

    def round(self, reqs: np.ndarray, arena: np.ndarray):
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        return self._chk(self.L.orc_round(self.h, reqs.ctypes.data, len(reqs),
                                          arena.ctypes.data if arena is not None else None), "round")

    def run_rounds(self, reqs: np.ndarray, round_n: np.ndarray, arena: np.ndarray, prune_bytes: int = 0):
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        round_n = np.ascontiguousarray(round_n, dtype=np.uint32)
        return self._chk(self.L.orc_run_rounds(self.h, reqs.ctypes.data, round_n.ctypes.data,
                                               len(round_n), arena.ctypes.data, prune_bytes), "run_rounds")

    def run_rounds_mt(self, reqs: np.ndarray, round_n: np.ndarray, arena: np.ndarray, prune_bytes: int = 0,
                      max_seconds: float = 60.0):
        """the same stream on one pinned thread per server (CPU baseline only)"""
        reqs = np.ascontiguousarray(reqs, dtype=REQ_DTYPE)
        round_n = np.ascontiguousarray(round_n, dtype=np.uint32)
        return self._chk(self.L.orc_run_rounds_mt(self.h, reqs.ctypes.data, round_n.ctypes.data, len(round_n),
                                                  arena.ctypes.data, prune_bytes, max_seconds), "run_rounds_mt")

    @property
    def force_prunes(self): return int(self.L.orc_force_prune_count(self.h))

    @property
    def leader(self): return int(self.L.orc_leader(self.h))

    def log(self, r) -> OracleLog:
        lg = OracleLog(handle=self.L.orc_replica_log(self.h, r))
        lg._keep = self
        return lg

    def sid(self, r): return int(self.L.orc_replica_sid(self.h, r))
    def cid_bitmask(self, r): return int(self.L.orc_replica_cid_bitmask(self.h, r))
    def term(self, r): return self.sid(r) >> 9
    def highest_rec(self, r): return int(self.L.orc_replica_highest_rec(self.h, r))
    def apply_count(self, r): return int(self.L.orc_replica_apply_count(self.h, r))
    def apply_hash(self, r): return int(self.L.orc_replica_apply_hash(self.h, r))
    def store_count(self, r): return int(self.L.orc_replica_store_count(self.h, r))

    def apply_log(self, r) -> np.ndarray:
        n = u64(0)
        ptr = self.L.orc_replica_apply_log(self.h, r, C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=APPLY_DTYPE)
        buf = (C.c_char * (n.value * APPLY_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=APPLY_DTYPE).copy()

    def round_record(self):
        n = int(self.L.orc_round_count(self.h))
        if n == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
        c = np.ctypeslib.as_array(self.L.orc_round_commit(self.h), shape=(n,)).copy()
        e = np.ctypeslib.as_array(self.L.orc_round_end(self.h), shape=(n,)).copy()
        return c, e


def fill_payload(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_fill_payload(seed, out.ctypes.data, n)
    return out


def canon(ring: np.ndarray, end: int, frm: int, to: int) -> tuple:
    """Canonical serialisation of the entries in [frm, to); returns (bytes, n_entries)."""
    ring = np.ascontiguousarray(ring, dtype=np.uint8)
    n = u64(0)
    need = lib().orc_canon(ring.ctypes.data, len(ring), end, frm, to, None, 0, C.byref(n))
    out = np.zeros(max(int(need), 1), dtype=np.uint8)
    lib().orc_canon(ring.ctypes.data, len(ring), end, frm, to, out.ctypes.data, int(need), C.byref(n))
    return out[:int(need)].tobytes(), int(n.value)


def canon_hash(ring: np.ndarray, end: int, frm: int, to: int) -> tuple:
    ring = np.ascontiguousarray(ring, dtype=np.uint8)
    n = u64(0)
    h = lib().orc_canon_hash(ring.ctypes.data, len(ring), end, frm, to, C.byref(n))
    return int(h), int(n.value)


def run_trace(trace, record_apply: bool = True, allow_exact_fit: bool = True,
              on_event=None) -> "Cluster":
    """Drive a fresh oracle cluster with an apus_amd.trace.Trace (duck-typed).
    on_event(i, event, cluster) is called after each event (used to snapshot
    quiescent points)."""
    c = Cluster(trace.group_size, trace.log_len, record_apply, allow_exact_fit)
    reqs = np.ascontiguousarray(trace.reqs, dtype=REQ_DTYPE)
    arena = np.ascontiguousarray(trace.arena, dtype=np.uint8)
    for i, ev in enumerate(trace.events):
        op = ev[0]
        if op == "ROUND":
            _, g0, n = ev
            c.round(reqs[g0:g0 + n], arena)
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
            raise ValueError(f"unknown trace event {ev}")
        if on_event is not None:
            on_event(i, ev, c)
    return c


def defined_mask(ring: np.ndarray, end: int, frm: int, to: int) -> np.ndarray:
    """Boolean mask of the ring bytes that the entries in [frm, to) define."""
    ring = np.ascontiguousarray(ring, dtype=np.uint8)
    mask = np.zeros(len(ring), dtype=np.uint8)
    lib().orc_defined_mask(ring.ctypes.data, len(ring), end, frm, to, mask.ctypes.data)
    return mask.astype(bool)
