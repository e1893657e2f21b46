"""Throughput / latency of the replica kernels (apus_replica.h) on BASELINE configs[1]'s stream:
device-resident rounds (apus_gpu_rep_run) and host-fed requests (the pinned multi-producer ring).
  python tools/rep_bench.py [--sweep] [--entries N] [--steps K]
Prints one JSON line per measurement."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from apus_amd import trace as T  # noqa: E402
from apus_amd.engine import Engine  # noqa: E402


def step_cmds(tr, eng):
    out, ev, i = [], tr.events, 0
    while i < len(ev):
        if ev[i][0] == "ROUND":
            j = i
            while j < len(ev) and ev[j][0] == "ROUND":
                j += 1
            out.append(("run", eng.round_of_g0[ev[i][1]], j - i))
            i = j
            continue
        if ev[i][0] == "PRUNE":
            out.append(("prune",))
        i += 1
    return out


def staged(n_rep, entries, payload, batch, steps, n_append, n_fwork, warmup=1):
    pb = int(os.environ.get("SWEEP_PRUNE_BYTES", "0")) or None
    tr = T.steady_trace(n_rep, entries, payload, 16, batch, log_len=T.DEFAULT_LOG, prune_bytes=pb)
    eng = Engine(n_rep, tr.log_len)
    try:
        eng.stage_trace(tr)
        eng.elect(0)
        eng.sync()
        cmds = step_cmds(tr, eng)
        eng.rep_start(idle_ms=5000, peer_ms=1000, n_append=n_append, n_fwork=n_fwork)

        one_by_one = bool(os.environ.get("SWEEP_ONE_BY_ONE"))

        def step(times=1):
            if not one_by_one:
                eng.rep_cmds(cmds, times)
                return
            for _ in range(times):
                for c in cmds:
                    if c[0] == "run":
                        eng.rep_run(c[1], c[2])
                    else:
                        eng.rep_prune()
        step(warmup)
        eng.rep_drain(timeout_ms=60000)
        t0 = time.perf_counter()
        step(steps)
        t_issue = time.perf_counter() - t0
        eng.rep_drain(timeout_ms=120000)
        dt = time.perf_counter() - t0
        st = eng.rep_stats()
        code = eng.rep_park()
        lat = eng.rep_latency_ns()
        lat_a = eng.rep_latency_appended_ns()
        roles = eng.rep_role_stats()
        eng.quiesce()
        total = (warmup + steps) * len(tr.reqs)
        ok = eng.status() == 0 and code == 0
        for r in range(n_rep):
            o = eng.offsets(r)
            ok = ok and (o["commit"] == o["end"] == o["apply"])
        ok = ok and eng.counters(0)["highest_rec"] == total
        return {"mode": "staged", "replicas": n_rep, "payload": payload, "n_append": n_append, "n_fwork": n_fwork,
                "entries_per_s": len(tr.reqs) * steps / dt, "ms_per_step": dt / steps * 1e3, "verified": bool(ok), "host_issue_frac": t_issue / dt,
                "entries_total": total, "launch_ms": eng.rep_launch_ms(),
                "exit": code, "status": eng.status_names(), "stats": st,
                "lat_us_p50": float(np.percentile(lat, 50)) / 1e3 if len(lat) else None,
                "lat_appended_us_p50": float(np.percentile(lat_a, 50)) / 1e3 if len(lat_a) else None, "roles": roles}
    finally:
        eng.close()


def hostfed(n_rep, payload, seconds, n_append, n_fwork, blk=4096, threads=(1, 2, 4, 8)):
    tr = T.steady_trace(n_rep, 1 << 16, payload, 16, 64, log_len=T.DEFAULT_LOG)
    eng = Engine(n_rep, tr.log_len)
    try:
        eng.elect(0)
        eng.sync()
        eng.rep_start(idle_ms=5000, peer_ms=1000, n_append=n_append, n_fwork=n_fwork)
        reqs = np.ascontiguousarray(tr.reqs[16:16 + blk])
        # latency first: one 64-entry round at a time, host submit -> highest_rec
        hl = eng.rep_roundtrip_ns(reqs[:64], tr.arena, 300) / 1e3
        hl1 = eng.rep_roundtrip_ns(reqs[:1], tr.arena, 300) / 1e3
        eng.rep_drain()
        code0 = eng.rep_park()                   # (the latency samples of the lone rounds: read before the bulk runs)
        lat = eng.rep_latency_ns()
        lat_a = eng.rep_latency_appended_ns()
        out = {"mode": "host-fed", "replicas": n_rep, "payload": payload,
               "host_rt64_us_p50": float(np.percentile(hl[20:], 50)), "host_rt1_us_p50": float(np.percentile(hl1[20:], 50)),
               "dev_seq_to_applied_us_p50": float(np.percentile(lat, 50)) / 1e3 if len(lat) else None,
               "dev_appended_to_applied_us_p50": float(np.percentile(lat_a, 50)) / 1e3 if len(lat_a) else None, "by_threads": {}}
        ok = code0 == 0
        for nt in threads:
            eng.rep_start(idle_ms=5000, peer_ms=1000, n_append=n_append, n_fwork=n_fwork)
            hr0 = eng.rep_highest_rec()
            n, dt = eng.rep_feed(reqs, tr.arena, nt, seconds, prune_every_reqs=(8 << 20) // (64 + payload))
            good = eng.rep_highest_rec() == hr0 + n
            code = eng.rep_park()
            out["by_threads"][str(nt)] = {"entries_per_s": n / dt, "verified": bool(good and code == 0)}
            if os.environ.get("APUS_REP_DBG"):
                rs = eng.rep_role_stats()
                out["by_threads"][str(nt)]["seq"] = {k: rs.get("sequencer", {}).get(k) for k in ("passes", "moved", "rounds", "us", "busy_us", "pcie_us", "pcie_polls", "flow_us")}
            ok = ok and good and code == 0
        eng.quiesce()
        for r in range(n_rep):
            o = eng.offsets(r)
            ok = ok and (o["commit"] == o["end"] == o["apply"])
        out["verified"] = bool(ok); out["status"] = eng.status_names()
        return out
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entries", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--n-append", type=int, default=0)
    ap.add_argument("--n-fwork", type=int, default=0)
    ap.add_argument("--replicas", type=int, default=3)
    ap.add_argument("--no-hostfed", action="store_true")
    ap.add_argument("--grid", default="", help="n_append:n_fwork,... for --replicas")
    ap.add_argument("--brief", action="store_true")
    a = ap.parse_args()
    combos = [(a.replicas, a.n_append, a.n_fwork)]
    if a.grid:
        combos = [(a.replicas, int(x.split(":")[0]), int(x.split(":")[1])) for x in a.grid.split(",")]
    elif a.sweep:
        combos = [(3, 0, 0), (3, 48, 48), (1, 0, 0), (5, 0, 0), (7, 0, 0)]
    for n_rep, na, nf in combos:
        try:
            res = staged(n_rep, a.entries, 64, 64, a.steps, na, nf)
            if a.brief:
                res = {k: res[k] for k in ("replicas", "n_append", "n_fwork", "entries_per_s", "verified", "entries_total", "launch_ms", "lat_us_p50", "lat_appended_us_p50", "roles")}
            print(json.dumps(res), flush=True)
        except Exception as exc:
            print(json.dumps({"mode": "staged", "replicas": n_rep, "n_append": na, "n_fwork": nf, "error": repr(exc)[:600]}), flush=True)
    if not a.no_hostfed:
        try:
            print(json.dumps(hostfed(a.replicas, 64, 1.0, a.n_append, a.n_fwork)), flush=True)
        except Exception as exc:
            print(json.dumps({"mode": "host-fed", "error": repr(exc)[:600]}), flush=True)


if __name__ == "__main__":
    main()
