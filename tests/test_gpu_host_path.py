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
    os.environ.update(server_idx="0", group_size="3", APUS_GPU_LOG_LEN=str(LOG), APUS_PRUNE_PERIOD_MS="100000000")
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

    eng = Engine.from_handle(L.apus_gpu_global(), 3, LOG)
    eng.quiesce()
    assert eng.status() == 0
    for r in range(3):
        compare_replica(eng, cl, r, tag="proxy path")
    L.apus_proxy_shutdown(p)


def test_concurrent_submitters_all_commit():
    """memcached-style: several application threads block in proxy_on_read at once."""
    from apus_amd import _lib
    L = _lib.load(build_if_missing=False)
    if not L.apus_gpu_global():
        os.environ.update(server_idx="0", group_size="3", APUS_GPU_LOG_LEN=str(1 << 17), APUS_PRUNE_PERIOD_MS="20")
        L.proxy_init.restype = C.c_void_p
        L.proxy_init.argtypes = [C.c_char_p, C.c_char_p]
        p = L.proxy_init(b"", None)
    else:
        pytest.skip("one SMR instance per process (global singleton, like the reference)")
