"""The drop-in host path on a GPU: the reference's proxy_* API (rsm-interface.h)
driven like the LD_PRELOAD hooks would drive it, through libapus_gpu.so's C host
layer -> live submission -> HIP kernels, checked against the oracle."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from apus_amd import trace as T

pytestmark = pytest.mark.gpu


def test_proxy_api_single_app_thread_matches_oracle():
    from apus_amd import _lib
    from apus_amd.engine import Engine
    from oracle import oracle as orc
    from tests.parity import compare_replica
    L = _lib.load(build_if_missing=False)
    LOG = 1 << 20
    os.environ.update(server_idx="0", group_size="3", APUS_GPU_LOG_LEN=str(LOG), APUS_PRUNE_PERIOD_MS="100000000",
                      APUS_PROXY_KEEP_ENGINE="1")
    L.proxy_init.restype = C.c_void_p
    L.proxy_init.argtypes = [C.c_char_p, C.c_char_p]
    for f in ("proxy_on_accept", "proxy_on_close"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        getattr(L, f).restype = None
    L.proxy_on_read.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int]
    L.proxy_on_read.restype = None
    L.apus_proxy_highest_rec.argtypes = [C.c_void_p]
    L.apus_proxy_highest_rec.restype = C.c_uint64
    L.apus_proxy_shutdown.argtypes = [C.c_void_p]
    L.is_leader.restype = C.c_int
    p = L.proxy_init(b"", None)
    assert p, "proxy_init failed"
    assert L.is_leader() == 1

    # the request stream an application thread would generate: 4 connections,
    # interleaved reads of various sizes, then closes -- every call blocks until
    # its entry is applied (proxy.c:160)
    rng = np.random.default_rng(5)
    types, fds, lens, bufs = [], [], [], []
    for fd in range(100, 104):
        types.append(T.CONNECT); fds.append(fd); lens.append(0); bufs.append(b"")
    for k in range(600):
        n = int(rng.choice([1, 40, 64, 107, 300, 1024]))
        types.append(T.SEND); fds.append(100 + k % 4); lens.append(n)
        bufs.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    for fd in range(100, 104):
        types.append(T.CLOSE); fds.append(fd); lens.append(0); bufs.append(b"")
    for t, fd, b in zip(types, fds, bufs):
        if t == T.CONNECT:
            L.proxy_on_accept(p, fd)
        elif t == T.SEND:
            buf = C.create_string_buffer(b, len(b))
            L.proxy_on_read(p, buf, len(b), fd)
        else:
            L.proxy_on_close(p, fd)
    assert L.apus_proxy_highest_rec(p) == len(types)

    # oracle: same admitted requests (ids from the same admission rule), one round each
    adm = T.Admission(0)
    reqs = np.zeros(len(types), dtype=orc.REQ_DTYPE)
    arena = bytearray(16)
    for g, (t, fd, b) in enumerate(zip(types, fds, bufs)):
        cid, rid = adm.connect(fd) if t == T.CONNECT else adm.send(fd) if t == T.SEND else adm.close(fd)
        reqs[g] = (rid, len(arena), cid, len(b), t, (0, 0, 0))
        arena += b + bytes((-len(b)) % 16)
    arena = np.frombuffer(bytes(arena) + bytes(32), dtype=np.uint8)
    cl = orc.Cluster(3, LOG)
    cl.elect(0)
    for g in range(len(reqs)):
        cl.round(reqs[g:g + 1], arena)
    cl.quiesce()

    # the DARE thread stops the persistent kernel and leaves the engine to us (APUS_PROXY_KEEP_ENGINE)
    L.apus_proxy_shutdown(p)
    L.apus_gpu_global.restype = C.c_void_p
    L.apus_gpu_destroy.argtypes = [C.c_void_p]
    eng = Engine.from_handle(L.apus_gpu_global(), 3, LOG)
    eng.leader, eng.term = 0, 2
    eng.quiesce()
    assert eng.status() == 0
    for r in range(3):
        compare_replica(eng, cl, r, tag="proxy path")
    L.apus_gpu_destroy(L.apus_gpu_global())


def test_concurrent_submitters_all_commit():
    """memcached-style: 8 application threads x 2000 proxy_on_read block at once; the leader's log is
    read back, replayed through the oracle in that order and every replica compared bit for bit
    (tests/_proxy_mt_worker.py; a process of its own: one SMR instance per process)."""
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(tempfile.mkdtemp(), "mt.json")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "_proxy_mt_worker.py"), out, "8", "2000"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert os.path.exists(out), f"worker died\n{p.stdout[-1500:]}\n{p.stderr[-3000:]}"
    res = json.load(open(out))
    assert res["ok"], f"{res.get('error')}\n{p.stderr[-1500:]}"
    assert res["total"] == 8 * 2001 and res["connections"] == 8
