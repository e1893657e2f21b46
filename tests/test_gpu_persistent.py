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
