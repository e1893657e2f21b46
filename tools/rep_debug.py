"""debug helper: run a catalogue trace through the replica kernels repeatedly, dump control blocks on a mismatch"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apus_amd.engine import Engine
from oracle import oracle as orc
from tests import traces
from tests.parity import compare_replica

name = sys.argv[1] if len(sys.argv) > 1 else "c3_small"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
source = sys.argv[3] if len(sys.argv) > 3 else "staged"
tr = {**traces.CATALOGUE, **traces.EXTRA}[name]()
cl = orc.run_trace(tr)
for it in range(reps):
    eng = Engine(tr.group_size, tr.log_len, capacity=max(tr.group_size, cl.n))
    try:
        eng.run_trace_rep(tr, source=source)
        bad = []
        for r in range(eng.group_size):
            w = eng.hdr_words(r)
            if int(w[7]) != tr.log_len:
                bad.append(r)
        if bad:
            print(f"iteration {it}: len word clobbered on replicas {bad}")
            for r in range(eng.group_size):
                print(r, [int(x) for x in eng.hdr_words(r)[:40]])
        eng.quiesce()
        for r in range(eng.group_size):
            try:
                compare_replica(eng, cl, r, tag=f"it {it}")
            except AssertionError as e:
                print(f"iteration {it}:", str(e)[:600])
    finally:
        eng.close()
print("done")
