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


@pytest.mark.parametrize("threads,block,lens", [(4, 1000, (64, 40)), (3, 4096, (64,)), (5, 257, (17, 64, 96, 300))])
def test_concurrent_producers_with_odd_blocks_and_mixed_lengths(threads, block, lens):
    """Round 6: the request ring's WINDOW words (RepReq.ready_win: one word per aligned window of 64 slots, written by whoever
    completes the window with requests of one length; producers that share a window settle it through counts in host memory,
    apus_engine.hip: rep_win_note) under what they were built for and what they must survive: several producer threads
    (apus_gpu_rep_submit, side by side) submitting the same block over and over -- block sizes that are not multiples of 64 (every block starts
    inside a window the producer before it began), lengths that change inside windows (no word: those go round by round), payloads
    too long for a slot (the arena path), one length throughout (every window gets its word).  The order in which concurrent
    producers' requests enter the log is theirs; checked is everything that does not depend on it: every request was committed and
    applied on every replica (highest_rec, offsets), the replicas are bit-identical -- and an ORACLE REPLAY of the leader's log in
    the order it has (tests/_cluster.py: oracle_replay) gives the same rings and offsets bit for bit.
    Reference: leader_handle_submit_req + get_tailq_message, src/proxy/proxy.c:108-161, dare_ibv_ud.c:780-790."""
    from apus_amd.engine import Engine
    from tests import _cluster as K
    n_rep, L = 3, T.DEFAULT_LOG
    rng = np.random.default_rng(7)
    conns = 8
    lens_req = np.asarray(lens, dtype=np.int64)[rng.integers(0, len(lens), block)]
    types = np.concatenate([np.full(conns, T.CONNECT), np.full(block, T.SEND)]).astype(np.uint8)
    fds = np.concatenate([np.arange(conns), np.arange(block) % conns]) + 100
    reqs, arena, _ = T.build_requests(types, fds, np.concatenate([np.zeros(conns, dtype=np.int64), lens_req]), 0)
    blk = np.ascontiguousarray(reqs[conns:])
    eng = Engine(n_rep, L)
    try:
        eng.elect(0)
        eng.sync()
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        hr0 = eng.rep_highest_rec()
        passes, errs = 6, []

        def producer():
            try:
                for _ in range(passes):
                    eng.rep_submit(blk, arena)               # (ctypes drops the GIL: the producers really run side by side)
            except Exception as exc:                         # noqa: BLE001
                errs.append(repr(exc))
        th = [threading.Thread(target=producer) for _ in range(threads)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        assert not errs, errs
        n = threads * passes * block
        eng.rep_drain(timeout_ms=20000)
        assert eng.rep_highest_rec() == hr0 + n, (n, eng.rep_highest_rec(), eng.rep_stats())
        assert eng.rep_park() == 0
        eng.quiesce()
        assert eng.status() == 0, eng.status_names()
        o0 = eng.offsets(0)
        assert o0["commit"] == o0["end"] == o0["apply"] and o0["end"] < L and o0["head"] == 0, o0   # (one lap, no prune tick: the replay below starts at offset 0)
        rings = {r: eng.ring(r) for r in range(n_rep)}
        reps = {r: eng.offsets(r) for r in range(n_rep)}
        for r in (1, 2):
            assert (reps[r]["commit"], reps[r]["end"], reps[r]["apply"]) == (o0["commit"], o0["end"], o0["apply"]), (r, reps[r], o0)
        ents = K.log_entries(rings[0], o0["end"])
        sends = [e for e in ents if e[2] == T.SEND]
        assert len(sends) == n and [e[0] for e in ents] == list(range(1, len(ents) + 1))
        # every producer's pass over the block is in the log in the block's order (a producer's own requests never overtake each other)
        cl = K.oracle_replay(n_rep, L, ents, [])
        K.compare_with_oracle(cl, reps, rings, [0, 1, 2])
    finally:
        eng.close()
