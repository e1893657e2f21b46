"""host-fed sanity check of the request ring's admission (one short GPU call): 4 producer threads for 0.3 s, every entry applied"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apus_amd import trace as T
from apus_amd.engine import Engine
tr = T.steady_trace(3, 1 << 13, 64, 16, 64, log_len=T.DEFAULT_LOG)
eng = Engine(3, tr.log_len)
try:
    eng.elect(0); eng.sync()
    out = {}
    for nt in (4, 1):
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        hr0 = eng.rep_highest_rec()
        n, sec = eng.rep_feed(np.ascontiguousarray(tr.reqs[16:16 + 4096]), tr.arena, nt, 0.3, prune_every_reqs=(8 << 20) // 128)
        ok = eng.rep_highest_rec() == hr0 + n
        code = eng.rep_park()
        out[str(nt)] = {"Meps": round(n / sec / 1e6, 1), "ok": bool(ok and code == 0)}
    eng.quiesce()
    o = [eng.offsets(r) for r in range(3)]
    out["settled"] = all(x["commit"] == x["end"] == x["apply"] for x in o) and eng.status() == 0
    print(json.dumps(out))
finally:
    eng.close()
