"""Launch-to-launch spread of the replica kernels on configs[1]: N launches of the same binary in ONE process, each with the XCD
(XCC_ID) map of its workgroups -- where the control workgroup and the followers' first workgroups (retire / apply wavefronts) ran,
how the append workgroups spread over the eight XCDs.
  python tools/rep_spread.py [replicas] [launches] [steps]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from apus_amd import trace as T  # noqa: E402
from apus_amd.engine import Engine  # noqa: E402
from rep_bench import step_cmds  # noqa: E402


def main():
    n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    tr = T.steady_trace(n_rep, 1 << 20, 64, 16, 64, log_len=T.DEFAULT_LOG)
    eng = Engine(n_rep, tr.log_len)
    vals = []
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        cmds = step_cmds(tr, eng)
        eng.L.apus_gpu_rep_xcc_map.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        import time
        for i in range(launches):
            eng.rep_start(idle_ms=5000, peer_ms=1000)
            eng.rep_cmds(cmds, 1)
            eng.rep_drain(timeout_ms=60000)
            t0 = time.perf_counter()
            eng.rep_cmds(cmds, steps)
            eng.rep_drain(timeout_ms=120000)
            dt = time.perf_counter() - t0
            code = eng.rep_park()
            xm = (C.c_uint8 * 1024)()
            eng.L.apus_gpu_rep_xcc_map(eng.h, xm)
            x = np.array(xm[:], dtype=np.int32)
            used = x[x > 0] - 1
            geps = len(tr.reqs) * steps / dt / 1e9
            vals.append(geps)
            hist = np.bincount(used, minlength=8).tolist()
            print(json.dumps({"launch": i, "G_entries_per_s": round(geps, 3), "exit": code, "workgroups": int(len(used)), "control_wg_xcd": int(x[0] - 1),
                              "round_robin": bool(np.all(used == (np.arange(len(used)) % 8 + used[0]) % 8)), "wgs_per_xcd": hist}), flush=True)
        eng.quiesce()
        ok = eng.status() == 0 and all(eng.offsets(r)["commit"] == eng.offsets(r)["end"] == eng.offsets(r)["apply"] for r in range(n_rep))
        v = np.array(vals)
        print(json.dumps({"replicas": n_rep, "launches": launches, "steps_per_launch": steps, "verified": bool(ok), "min": float(v.min()), "median": float(np.median(v)),
                          "max": float(v.max()), "spread_pct": float((v.max() - v.min()) / np.median(v) * 100)}), flush=True)
    finally:
        eng.close()


if __name__ == "__main__":
    main()
