#!/usr/bin/env python3
"""Where an append wavefront's and a follower work wavefront's round goes (APUS_REP_DBG=256: in-kernel phase timers), per
configuration of the bench line:  python tools/rep_phase.py CFG [CFG ...]   (CFG as in rep_profile_run.py)
The timers cost throughput; the split is what matters."""
import json
import os
import sys

os.environ["APUS_REP_DBG"] = os.environ.get("APUS_REP_DBG", "256")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import time  # noqa: E402

from apus_amd.engine import Engine  # noqa: E402
from rep_bench import step_cmds  # noqa: E402
from rep_profile_run import make_trace  # noqa: E402

for cfg in sys.argv[1:]:
    tr = make_trace(cfg)
    eng = Engine(tr.group_size, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        cmds = step_cmds(tr, eng)
        eng.rep_start(idle_ms=5000, peer_ms=1000)
        t0 = time.perf_counter()
        for _ in range(3):
            for c in cmds:
                eng.rep_run(c[1], c[2]) if c[0] == "run" else eng.rep_prune()
        eng.rep_drain(timeout_ms=120000)
        dt = time.perf_counter() - t0
        eng.rep_park()
        r = eng.rep_role_stats()
        out = {"cfg": cfg, "G_entries_per_s": round(3 * len(tr.reqs) / dt / 1e9, 3)}
        for k in ("append", "f0_work"):
            if k in r:
                out[k] = {a: round(b, 2) for a, b in r[k].items()}
        print(json.dumps(out), flush=True)
    finally:
        eng.close()
