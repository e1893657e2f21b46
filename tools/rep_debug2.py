"""debug helper: several traces in one process (engine after engine), looking for clobbered control words"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apus_amd import trace as T
from apus_amd.engine import Engine
from oracle import oracle as orc
from tests import traces
from tests.parity import compare_replica

seq = [("mixed5", T.steady_trace(5, 1500, (40, 64, 107, 1024, 4096), 16, (1, 64), log_len=1 << 20, seed=3), "pinned", False),
       ("mixed5", T.steady_trace(5, 1500, (40, 64, 107, 1024, 4096), 16, (1, 64), log_len=1 << 20, seed=3), "staged", False),
       ("steady7", traces.steady7_mixed(), "pinned", True),
       ("c2_small", traces.c2_small(), "staged", False), ("c3_small", traces.c3_small(), "staged", False), ("c4_small", traces.c4_small(), "staged", False)]
cls = [orc.run_trace(t[1]) for t in seq]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for (name, tr, source, de), cl in zip(seq, cls):
        eng = Engine(tr.group_size, tr.log_len, capacity=max(tr.group_size, cl.n))
        try:
            w0 = [eng.hdr_words(r) for r in range(eng.group_size)]
            eng.run_trace_rep(tr, source=source, drain_each=de)
            for r in range(eng.group_size):
                w = eng.hdr_words(r)
                if int(w[7]) != tr.log_len:
                    print(f"it {it} {name}/{source}: replica {r} len word = {int(w[7])}; before the run it was {int(w0[r][7])}")
                    print("   now   ", [int(x) for x in w[:24]])
                    print("   before", [int(x) for x in w0[r][:24]])
            eng.quiesce()
        finally:
            eng.close()
print("done")
