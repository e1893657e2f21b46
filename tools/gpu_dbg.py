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

if __name__ == "__main__":
    globals()[sys.argv[1]]()
