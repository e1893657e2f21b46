"""scratch driver for one-off GPU diagnostics (kept small; see the call scripts under tools/)"""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apus_amd import trace as T
from apus_amd.engine import Engine

def c2_full():
    tr = T.config_c2()
    eng = Engine(3, tr.log_len)
    try:
        eng.run_trace_rep(tr, source="staged", idle_ms=20000, peer_ms=5000)
        print("c2 full ok", eng.offsets(0))
    except Exception:
        traceback.print_exc()
        try:
            print("stats", eng.rep_stats(), eng.status_names())
        except Exception as e:
            print("no stats", e)
    finally:
        eng.close()

def two_runs():
    """a pinned (lone-round) run, then a staged run on the same engine: what bench.py's replica measurement does"""
    import numpy as np
    tr = T.steady_trace(3, 1 << 18, 64, 16, 64, log_len=T.DEFAULT_LOG)
    eng = Engine(3, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        reqs64 = np.ascontiguousarray(tr.reqs[16:16 + 64])
        first = os.environ.get("FIRST", "pinned")
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        if first == "pinned":
            eng.rep_roundtrip_ns(reqs64, tr.arena, int(os.environ.get("ITERS", "300")))
        else:
            eng.rep_run(0, 100)
        eng.rep_drain()
        print("first run parked", eng.rep_park(), eng.status_names(), flush=True)
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        n_rounds = sum(1 for e in tr.events if e[0] == "ROUND")
        eng.rep_run(0, n_rounds)
        eng.rep_drain(timeout_ms=20000)
        print("second run parked", eng.rep_park(), eng.status_names(), eng.rep_stats(), flush=True)
    except Exception:
        traceback.print_exc()
    finally:
        eng.close()


if __name__ == "__main__":
    globals()[sys.argv[1]]()
