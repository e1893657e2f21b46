"""Worker of tests/test_gpu_host_path.py::test_concurrent_submitters_all_commit (one SMR instance per
process, like the reference: the test needs a process of its own).

N application threads block in proxy_on_read at once (memcached-style; the reference's
leader_handle_submit_req, proxy.c:108-161: id assignment and enqueue under the tailq spinlock, then a
spin on highest_rec).  The admission order across threads is not determined, so the check goes the
other way round: the leader's log IS the order; read it back, feed exactly that request sequence to
the oracle, and compare every replica bit for bit.  Per connection the entries must carry that
thread's payloads in the order it sent them, with req_id 1, 2, 3, ..."""
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def main():
    out_path, n_threads, per_thread = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from apus_amd import _lib, trace as T
    from apus_amd.engine import Engine
    from oracle import oracle as orc
    from tests.parity import compare_replica
    res = {"ok": False}
    try:
        L = _lib.load(build_if_missing=False)
        LOG = 1 << 24
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
        L.apus_gpu_global.restype = C.c_void_p
        L.apus_gpu_destroy.argtypes = [C.c_void_p]
        p = L.proxy_init(b"", None)
        assert p, "proxy_init failed"

        sizes = [1, 40, 64, 107, 300, 1024]
        sent = {}
        for t in range(n_threads):
            rng = np.random.default_rng(100 + t)
            sent[200 + t] = [rng.integers(0, 256, int(rng.choice(sizes)), dtype=np.uint8).tobytes() for _ in range(per_thread)]
        errors = []

        def app_thread(fd):
            try:
                L.proxy_on_accept(p, fd)
                for b in sent[fd]:
                    buf = C.create_string_buffer(b, len(b))
                    L.proxy_on_read(p, buf, len(b), fd)       # returns when the entry is applied (proxy.c:160)
            except BaseException as e:      # noqa: BLE001
                errors.append(repr(e))

        th = [threading.Thread(target=app_thread, args=(200 + t,)) for t in range(n_threads)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=240)
        assert not any(t.is_alive() for t in th), "a submitter is still blocked"
        assert not errors, errors
        total = n_threads * (per_thread + 1)
        hr = int(L.apus_proxy_highest_rec(p))
        assert hr == total, f"highest_rec {hr} != {total}"
        L.apus_proxy_shutdown(p)                              # stops the persistent kernel; the engine stays (KEEP_ENGINE)

        eng = Engine.from_handle(L.apus_gpu_global(), 3, LOG)
        eng.leader, eng.term = 0, 2
        eng.quiesce()
        assert eng.status() == 0, eng.status_names()
        # the leader's log, in order
        recs = eng.apply_records(0, 0, eng.counters(0)["n_apply"])       # slot 0 = the blank CONFIG entry
        recs = recs[recs["kind"] == 1]
        assert len(recs) == total, f"{len(recs)} client entries in the log, {total} submitted"
        ring = eng.ring(0)
        reqs = np.zeros(total, dtype=orc.REQ_DTYPE)
        arena = bytearray(16)
        seen = {}
        conn_of = {}
        for g, r in enumerate(recs):
            off, ln, cid, typ = int(r["off"]), int(r["len"]), int(r["clt_id"]), int(r["type"])
            rid = int(ring[off + 16:off + 24].view(np.uint64)[0])
            k = seen.get(cid, 0)
            assert rid == k + 1, f"connection {cid:#x}: req_id {rid} after {k}"
            seen[cid] = rid
            body = ring[off + 50:off + 50 + ln].tobytes()
            if typ == T.CONNECT:
                assert ln == 0 and k == 0
            else:
                assert typ == T.SEND
                conn_of.setdefault(cid, []).append(body)
            reqs[g] = (rid, len(arena), cid, ln, typ, (0, 0, 0))
            arena += body + bytes((-ln) % 16)
        # per connection: one thread's payloads, in the order it sent them
        got = sorted(conn_of.values(), key=lambda v: v[0])
        want = sorted(sent.values(), key=lambda v: v[0])
        assert len(conn_of) == n_threads and got == want, "a connection's entries are not what its thread sent, in order"
        arena = np.frombuffer(bytes(arena) + bytes(32), dtype=np.uint8)
        cl = orc.Cluster(3, LOG)
        cl.elect(0)
        for g in range(total):
            cl.round(reqs[g:g + 1], arena)
        cl.quiesce()
        for r in range(3):
            compare_replica(eng, cl, r, tag="threaded proxy path")
        L.apus_gpu_destroy(L.apus_gpu_global())
        res.update(ok=True, total=total, connections=len(conn_of))
    except BaseException as e:      # noqa: BLE001
        import traceback
        res["error"] = repr(e) + "\n" + traceback.format_exc()[-1500:]
    json.dump(res, open(out_path, "w"))
    os._exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
