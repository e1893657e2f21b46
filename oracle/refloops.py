"""ctypes face of the reference-as-is cluster (oracle/_ref/libapus_fabric.so +
libapus_ref_loops.so): N instances of the UNMODIFIED reference server sources
(/root/reference/src/dare/*.c) behind the in-process verbs stand-in of oracle/refshim/.

TEST INFRASTRUCTURE ONLY.  It exists to pin the restated oracle (oracle/apus_oracle.c)
to outputs of the reference itself: tests/test_oracle_vs_refloops.py replays the same
traces on both, tests/golden/make_cluster_golden.py writes the committed fixtures.
`RefCluster` duck-types `oracle.Cluster`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from . import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
FABRIC_SO = os.path.join(HERE, "_ref", "libapus_fabric.so")
LOOPS_SO = os.path.join(HERE, "_ref", "libapus_ref_loops.so")
REF_SRC = "/root/reference/src/dare/dare_server.c"

u64, u32, u16, u8 = C.c_uint64, C.c_uint32, C.c_uint16, C.c_uint8
vp = C.c_void_p

REF_APPLY_DTYPE = np.dtype([("off", "<u8"), ("idx", "<u8"), ("len", "<u4"), ("clt_id", "<u2"),
                            ("type", "u1"), ("kind", "u1")])
assert REF_APPLY_DTYPE.itemsize == 24

# the reference's own sample settings (target/nodes.local.cfg:20-35, "DEBUG" block)
CFG_TEXT = """# written by oracle/refloops.py: values of the reference's target/nodes.local.cfg
dare_global_config = {
    hb_period = 0.01;
    elec_timeout_low = 100000;
    elec_timeout_high = 300000;
    retransmit_period = 0.04;
    rc_info_period = 0.05;
    log_pruning_period = 0.05;
};
"""

_lib = None


def build(force: bool = False) -> None:
    if not os.path.exists(REF_SRC):
        return
    srcs = [os.path.join(HERE, "refshim", f) for f in ("fabric.c", "refcluster.c", "glue.c", "glue_store.c", "fabric.h",
                                                          "refcluster.h", "ev.h", "libconfig.h",
                                                          os.path.join("infiniband", "verbs.h"))]
    stale = force or not (os.path.exists(FABRIC_SO) and os.path.exists(LOOPS_SO))
    if not stale:
        t = min(os.path.getmtime(FABRIC_SO), os.path.getmtime(LOOPS_SO))
        stale = any(os.path.getmtime(s) > t for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", HERE, "loops"], stdout=subprocess.DEVNULL)


def available() -> bool:
    build()
    return os.path.exists(FABRIC_SO) and os.path.exists(LOOPS_SO)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libapus_ref_loops.so is not built and /root/reference is absent")
        L = C.CDLL(FABRIC_SO, mode=C.RTLD_GLOBAL)     # the instance copies resolve ibv_* against it
        pu64 = C.POINTER(u64)
        sig = {
            "refc_new": (vp, [C.c_int, u64, C.c_char_p, C.c_char_p, C.c_char_p]),
            "refc_free": (None, [vp]),
            "refc_error": (C.c_char_p, [vp]),
            "refc_elect": (C.c_int, [vp, C.c_int]),
            "refc_round": (C.c_int, [vp, vp, C.c_int, vp]),
            "refc_tick_prune": (C.c_int, [vp]),
            "refc_kill": (C.c_int, [vp, C.c_int]),
            "refc_hold": (C.c_int, [vp, C.c_int]),
            "refc_release": (C.c_int, [vp, C.c_int]),
            "refc_quiesce": (C.c_int, [vp]),
            "refc_join": (C.c_int, [vp, C.c_int]),
            "refc_state": (u64, [vp, C.c_int]),
            "refc_poll": (C.c_int, [vp, C.c_int]),
            "refc_fire": (C.c_int, [vp, C.c_int, C.c_int]),
            "refc_leader": (C.c_int, [vp]),
            "refc_group_size": (C.c_int, [vp]),
            "refc_alive": (C.c_int, [vp, C.c_int]),
            "refc_gone": (C.c_int, [vp, C.c_int]),
            "refc_offsets": (None, [vp, C.c_int, pu64]),
            "refc_entries": (vp, [vp, C.c_int]),
            "refc_sid": (u64, [vp, C.c_int]),
            "refc_prev_head": (C.c_int, [vp, C.c_int]),
            "refc_highest_rec": (u64, [vp, C.c_int]),
            "refc_store_count": (u64, [vp, C.c_int]),
            "refc_apply_count": (u64, [vp, C.c_int]),
            "refc_record_apply": (None, [vp, C.c_int]),
            "refc_apply_log": (vp, [vp, C.c_int, pu64]),
            "refc_cid": (None, [vp, C.c_int, pu64]),
            "refc_record_store": (None, [vp, C.c_int]),
            "refc_store_stream": (vp, [vp, C.c_int, pu64]),
            "refc_records_len": (u32, [vp, C.c_int]),
            "refc_peer": (None, [vp, C.c_int, C.c_int, pu64]),
            "refc_round_count": (u64, [vp]),
            "refc_round_commit": (pu64, [vp]),
            "refc_round_end": (pu64, [vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class _RefLogView:
    """offsets()/ring() of one server's dare_log_t, same face as oracle._LogBase."""

    def __init__(self, cluster: "RefCluster", r: int):
        self.c = cluster
        self.r = r

    def offsets(self) -> dict:
        out = (u64 * 8)()
        self.c.L.refc_offsets(self.c.h, self.r, out)
        return dict(zip(("head", "apply", "commit", "end", "tail", "old_end", "old_commit", "len"),
                        [int(v) for v in out]))

    def ring(self) -> np.ndarray:
        n = self.offsets()["len"]
        ptr = self.c.L.refc_entries(self.c.h, self.r)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(u8)), shape=(n,))

    @property
    def prev_head(self) -> int:
        return int(self.c.L.refc_prev_head(self.c.h, self.r))


class RefCluster:
    """N unmodified reference servers driven by trace events (one live cluster at a time)."""

    _live = None

    def __init__(self, group_size: int, log_len: int = orc.DEFAULT_LOG, record_apply: bool = True,
                 log_dir: str | None = None):
        self.L = lib()
        if RefCluster._live is not None:
            raise RuntimeError("only one RefCluster at a time (the fabric is process-wide)")
        self._tmp = tempfile.TemporaryDirectory(prefix="apus_refloops_")
        cfg = os.path.join(self._tmp.name, "nodes.cfg")
        with open(cfg, "w") as f:
            f.write(CFG_TEXT)
        log_dir = log_dir or os.environ.get("APUS_REF_LOG_DIR", "")
        self.h = self.L.refc_new(group_size, log_len, LOOPS_SO.encode(), cfg.encode(), log_dir.encode())
        if not self.h:
            raise RuntimeError("refc_new failed")
        RefCluster._live = self
        self.n = group_size
        self.log_len = log_len
        self.L.refc_record_apply(self.h, int(record_apply))
        self.L.refc_record_store(self.h, int(record_apply))

    def close(self):
        if getattr(self, "h", None):
            self.L.refc_free(self.h)
            self.h = None
            RefCluster._live = None
            self._tmp.cleanup()

    def __del__(self):
        self.close()

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"reference cluster {what} failed rc={rc}: {self.L.refc_error(self.h).decode()}")
        return rc

    def elect(self, w): return self._chk(self.L.refc_elect(self.h, w), "elect")
    def tick_prune(self): return self._chk(self.L.refc_tick_prune(self.h), "tick_prune")
    def kill(self, r): return self._chk(self.L.refc_kill(self.h, r), "kill")
    def hold(self, r): return self._chk(self.L.refc_hold(self.h, r), "hold")
    def release(self, r): return self._chk(self.L.refc_release(self.h, r), "release")
    def quiesce(self): return self._chk(self.L.refc_quiesce(self.h), "quiesce")

    def join(self, r):
        rc = self._chk(self.L.refc_join(self.h, r), "join")
        self.n = int(self.L.refc_group_size(self.h))
        return rc
    def poll(self, r): return self.L.refc_poll(self.h, r)
    def fire(self, r, which): return self.L.refc_fire(self.h, r, which)

    def round(self, reqs: np.ndarray, arena: np.ndarray):
        reqs = np.ascontiguousarray(reqs, dtype=orc.REQ_DTYPE)
        return self._chk(self.L.refc_round(self.h, reqs.ctypes.data, len(reqs),
                                           arena.ctypes.data if arena is not None else None), "round")

    @property
    def leader(self): return int(self.L.refc_leader(self.h))
    def alive(self, r): return bool(self.L.refc_alive(self.h, r))
    def gone(self, r): return bool(self.L.refc_gone(self.h, r))
    def log(self, r) -> _RefLogView: return _RefLogView(self, r)
    def sid(self, r): return int(self.L.refc_sid(self.h, r))
    def term(self, r): return self.sid(r) >> 9
    def highest_rec(self, r): return int(self.L.refc_highest_rec(self.h, r))
    def apply_count(self, r): return int(self.L.refc_apply_count(self.h, r))
    def store_count(self, r): return int(self.L.refc_store_count(self.h, r))

    def cid(self, r) -> dict:
        out = (u64 * 4)()
        self.L.refc_cid(self.h, r, out)
        return {"epoch": int(out[0]), "size0": int(out[1]) & 0xFF, "size1": (int(out[1]) >> 8) & 0xFF,
                "state": (int(out[1]) >> 16) & 0xFF, "bitmask": int(out[2]), "cid_offset": int(out[3])}

    def peer(self, r, i) -> dict:
        out = (u64 * 6)()
        self.L.refc_peer(self.h, r, i, out)
        return {"step": int(out[0]), "send_flag": int(out[1]), "fail_count": int(out[2]),
                "end": int(out[3]), "commit": int(out[4]), "connected": bool(out[5] & 1),
                "log_access": bool(out[5] & 2), "vote_ack": bool(out[5] & 4)}

    def apply_log(self, r) -> np.ndarray:
        n = u64(0)
        ptr = self.L.refc_apply_log(self.h, r, C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=REF_APPLY_DTYPE)
        buf = (C.c_char * (n.value * REF_APPLY_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=REF_APPLY_DTYPE).copy()

    def record_store(self, on=True): self.L.refc_record_store(self.h, int(on))

    def store_stream(self, r) -> bytes:
        n = u64(0)
        ptr = self.L.refc_store_stream(self.h, r, C.byref(n))
        return C.string_at(ptr, n.value) if n.value else b""

    def records_len(self, r): return int(self.L.refc_records_len(self.h, r))

    def round_record(self):
        n = int(self.L.refc_round_count(self.h))
        if n == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
        c = np.ctypeslib.as_array(self.L.refc_round_commit(self.h), shape=(n,)).copy()
        e = np.ctypeslib.as_array(self.L.refc_round_end(self.h), shape=(n,)).copy()
        return c, e


def run_trace(trace, record_apply: bool = True, on_event=None) -> RefCluster:
    """Replay an apus_amd.trace.Trace on the reference itself (same loop as oracle.run_trace)."""
    c = RefCluster(trace.group_size, trace.log_len, record_apply)
    reqs = np.ascontiguousarray(trace.reqs, dtype=orc.REQ_DTYPE)
    arena = np.ascontiguousarray(trace.arena, dtype=np.uint8)
    try:
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
    except Exception:
        c.close()
        raise
    return c
