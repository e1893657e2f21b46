#!/usr/bin/env python3
"""ONE resident launch of the replica kernels over one configuration of the bench line, for a profiler to wrap:
  python tools/rep_profile_run.py CFG [--steps K]
CFG: c2x1 | c2x3 | c2x5 | c2x7 (BASELINE configs[1]'s stream at 1 / 3 / 5 / 7 logical replicas) | c3 | c4 (configs[2], [3]).
1 + K steps of the staged stream in one launch (warm-up step included: the counters see the whole launch), then the
park.  Prints one JSON line: what the launch committed, its duration by HIP events, verified."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from apus_amd import trace as T  # noqa: E402
from apus_amd.engine import Engine  # noqa: E402
from rep_bench import step_cmds  # noqa: E402

CFGS = ("c2x1", "c2x3", "c2x5", "c2x7", "c3", "c4")


def make_trace(cfg):
    if cfg.startswith("c2x"):
        g = int(cfg[3:])
        return T.steady_trace(g, 1 << 20, 64, 16, 64, log_len=T.DEFAULT_LOG, name=f"C2x{g}")
    return T.config_c3() if cfg == "c3" else T.config_c4()


def run(cfg, steps):
    tr = make_trace(cfg)
    n_rep = tr.group_size
    eng = Engine(n_rep, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        cmds = step_cmds(tr, eng)
        eng.rep_start(idle_ms=5000, peer_ms=1000)

        def step(times=1):
            eng.rep_cmds(cmds, times)
        step()
        eng.rep_drain(timeout_ms=60000)
        t0 = time.perf_counter()
        step(steps)
        eng.rep_drain(timeout_ms=120000)
        dt = time.perf_counter() - t0
        code = eng.rep_park()
        launch_ms = eng.rep_launch_ms()
        eng.quiesce()
        total = (1 + steps) * len(tr.reqs)
        ok = eng.status() == 0 and code == 0 and eng.counters(0)["highest_rec"] == total
        for r in range(n_rep):
            o = eng.offsets(r)
            ok = ok and (o["commit"] == o["end"] == o["apply"])
        return {"cfg": cfg, "replicas": n_rep, "steps": steps, "entries_per_step": len(tr.reqs), "entries_total": total,
                "mean_entry_bytes": 64.0 + float(np.mean(tr.reqs["len"])), "launch_ms": launch_ms,
                "entries_per_s": len(tr.reqs) * steps / dt, "entries_per_s_by_launch": total / (launch_ms / 1e3) if launch_ms else None,
                "verified": bool(ok)}
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", choices=CFGS)
    ap.add_argument("--steps", type=int, default=0)
    a = ap.parse_args()
    steps = a.steps or (8 if a.cfg.startswith("c2") else 6)
    print(json.dumps(run(a.cfg, steps)), flush=True)


if __name__ == "__main__":
    main()
