"""bisect helper: which ABI call of the force_log_pruning path fails (prints progress unbuffered)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apus_amd.engine import Engine
from tests import traces
tr = traces.evict_slow_follower()
eng = Engine(tr.group_size, tr.log_len)
eng.reset(); eng.stage_trace(tr)
def P(*a): print(*a, flush=True)
k = 0
for ev in tr.events:
    if ev[0] == "ELECT":
        eng.elect(ev[1]); P("elect ok"); P("fp", eng.force_prune())
    elif ev[0] == "ROUND":
        k += 1
        eng.run_rounds(eng.round_of_g0[ev[1]], 1); P("round", k, "launched")
        eng.sync(); P("round", k, "synced", eng.offsets(0)["end"])
        P("fp", eng.force_prune())
    elif ev[0] == "HOLD":
        eng.hold(ev[1]); P("hold")
    elif ev[0] == "QUIESCE":
        eng.quiesce(); P("q", eng.force_prune())
    if k > 70: break
P("done", eng.status_names())
