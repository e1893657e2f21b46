"""The persistent consensus kernel (one resident workgroup per replica, host command
ring, device doorbells) must produce exactly the same logs as the oracle."""
import numpy as np
import pytest

from apus_amd import trace as T

pytestmark = pytest.mark.gpu


def run_and_compare(tr, n, L):
    from apus_amd.engine import Engine
    from oracle import oracle as orc
    from tests.parity import compare_replica
    cl = orc.run_trace(tr)
    eng = Engine(n, L)
    try:
        eng.run_trace_persistent(tr, idle_ms=3000, peer_ms=500)
        eng.quiesce()
        assert eng.status() == 0
        for r in range(n):
            compare_replica(eng, cl, r, tag="persistent")
        return eng.persist_latency_ns()
    finally:
        eng.close()


def test_persistent_three_replicas_64B():
    tr = T.steady_trace(3, 3000, 64, 8, 64, log_len=1 << 16)
    lat = run_and_compare(tr, 3, 1 << 16)
    assert len(lat) > 0 and np.median(lat) < 5e6          # a round commits in well under 5 ms


def test_persistent_mixed_sizes_five_replicas():
    tr = T.steady_trace(5, 1500, (40, 64, 107, 1024, 4096), 16, (1, 64), log_len=1 << 20, seed=3)
    run_and_compare(tr, 5, 1 << 20)


def test_persistent_exits_on_idle_limit():
    from apus_amd.engine import Engine
    import time
    eng = Engine(3, 1 << 16)
    try:
        eng.elect(0)
        eng.persist_start(idle_ms=50, peer_ms=50)
        time.sleep(1.0)
        code = eng.persist_stop()
        assert code in (0, 1)
    finally:
        eng.close()


def test_persistent_kernel_refuses_a_round_that_does_not_fit():
    """the live loop's log-full rule (dare_log.h:168, 492-495): the leader workgroup refuses the whole
    round before it stores anything, raises LOG_FULL and tells the host (apus_gpu_persist_full);
    nothing of the round is in any log and highest_rec does not move"""
    from apus_amd.engine import Engine
    n, L = 3, 1 << 14
    tr = T.steady_trace(n, 400, 64, 4, 8, log_len=L)
    rounds = [e for e in tr.events if e[0] == "ROUND"]
    eng = Engine(n, L)
    try:
        eng.elect(0)
        eng.sync()
        eng.persist_start(idle_ms=3000, peer_ms=500)
        lib = eng.L
        lib.apus_gpu_persist_full.argtypes = [type(eng.h)]
        accepted, appended, refused_bytes = 0, 64, 0          # 64: the blank CONFIG entry of the election
        hr_before = eng.persist_highest_rec()
        for ev in rounds:
            rq = tr.reqs[ev[1]:ev[1] + ev[2]]
            nbytes = int((64 + rq["len"].astype(np.int64)).sum())
            eng.persist_submit(rq, tr.arena)
            eng.persist_drain()
            if lib.apus_gpu_persist_full(eng.h):
                refused_bytes = nbytes
                break
            accepted += 1
            appended += nbytes
            assert eng.persist_highest_rec() == hr_before + ev[2]
            hr_before = eng.persist_highest_rec()
        assert lib.apus_gpu_persist_full(eng.h) == 1, f"{accepted} rounds went into a 16 KiB ring"
        assert appended <= L < appended + refused_bytes      # refused exactly when it no longer fits (head = 0: nothing pruned)
        assert eng.persist_highest_rec() == hr_before
        eng.persist_stop()
        assert eng.status() & 2                   # APUS_ST_LOG_FULL
        eng.L.apus_gpu_clear_status(eng.h)
        for r in range(n):
            o = eng.offsets(r)
            assert o["end"] == appended, f"replica {r}: {o}"
    finally:
        eng.close()
