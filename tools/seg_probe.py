#!/usr/bin/env python3
"""How does a k_step launch scale with the number / size of its segments?  Same bytes per launch,
different call sizes (prune tick period): a launch bound by the segment-to-segment chain costs the
same per segment whatever the segment's size, a throughput-bound one the same per byte."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from apus_amd import trace as T
from apus_amd.engine import Engine


def run(prune_mib, batch=64, entries=1 << 20, steps=5):
    tr = T.steady_trace(3, entries, 64, 16, batch, log_len=T.DEFAULT_LOG, prune_bytes=int(prune_mib * (1 << 20)))
    eng = Engine(3, tr.log_len, device=0)
    eng.stage_trace(tr)
    eng.elect(0)
    calls = bench.step_calls(tr, eng)
    bench.issue(eng, calls); eng.sync(); eng.check_status()
    eng.set_timing(True)
    for _ in range(steps):
        bench.issue(eng, calls)
    eng.sync()
    k_ms, k_launches = eng.kernel_time(0)
    eng.set_timing(False)
    n_calls = sum(1 for c in calls if c[0] == "rounds")
    per_launch_us = k_ms * 1e3 / k_launches
    total_us = k_ms * 1e3 / steps
    print(f"prune every {prune_mib:5.2f} MiB, rounds of {batch}: {n_calls} calls/step, {k_launches // steps} launches/step, "
          f"{per_launch_us:7.2f} us/launch, {total_us:7.1f} us of kernels/step, {total_us / n_calls:6.2f} us per call", flush=True)
    eng.check_status()
    eng.close()


if __name__ == "__main__":
    for p in (8, 4, 2, 1, 16):
        run(p)
    run(8, batch=32)
    run(8, batch=16)
