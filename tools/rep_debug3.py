import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apus_amd import trace as T
from apus_amd.engine import Engine
from oracle import oracle as orc
from tests import traces
from tests.parity import compare_replica, compare_apply_tail

seq = [("c2_small", traces.c2_small(), "staged"), ("c3_small", traces.c3_small(), "staged"), ("c4_small", traces.c4_small(), "staged")]
cls = [orc.run_trace(t[1]) for t in seq]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for (name, tr, source), cl in zip(seq, cls):
        eng = Engine(tr.group_size, tr.log_len, capacity=max(tr.group_size, cl.n))
        try:
            eng.run_trace_rep(tr, source=source)
            eng.quiesce()
            for r in range(eng.group_size):
                o = eng.offsets(r)
                if o["len"] != tr.log_len:
                    o2 = eng.offsets(r)
                    w = eng.hdr_words(r)
                    print(f"it {it} {name}: replica {r} offsets() len={o['len']}; read again: {o2['len']}; hdr_words: {int(w[7])}")
                compare_replica(eng, cl, r, tag=name)
                compare_apply_tail(eng, cl, r)
        except AssertionError as e:
            print(f"it {it} {name}: {str(e)[:300]}")
        finally:
            eng.close()
print("done")
