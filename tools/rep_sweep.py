"""A sweep of the replica kernels over workgroup grids and measurement knobs in ONE process (one library):
  python tools/rep_sweep.py "label:replicas:n_append:n_fwork:dbg" ...      (dbg = APUS_REP_DBG, read at every rep_start)
One condensed line per run: entries/s, verified, lone-round latencies, per-role pass statistics, phase timers."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rep_bench import staged  # noqa: E402


def brief(d):
    r = d["roles"]

    def f(k):
        if k not in r:
            return None
        x = r[k]
        return (x["moved"], round(x["rounds"] / max(1, x["moved"])), round(x["busy_us"] / max(1, x["moved"]), 2), round(x["busy_us"] / max(1e-9, x["us"]), 2))
    out = {"Meps": round(d["entries_per_s"] / 1e6), "ok": d["verified"], "host": round(d.get("host_issue_frac", 0), 2), "lat": d["lat_us_p50"], "lat_app": d["lat_appended_us_p50"],
           "seq_prune_us": r.get("sequencer", {}).get("y_us"), "seq_x": r.get("sequencer", {}).get("x"), "seq_us": r.get("sequencer", {}).get("us"),
           "seq_more": {k: r.get("sequencer", {}).get(k) for k in ("outer_passes", "flow_us", "pcie_us", "pcie_polls", "reloads", "pass_phases_us", "prune_phases_us")},
           "seq": f("sequencer"), "com": f("committer"), "app": f("applier"), "f0r": f("f0_retire"), "f0a": f("f0_apply")}
    for k in ("append", "f0_work"):
        if k in r:
            out[k] = {a: round(b, 2) for a, b in r[k].items()}
    return out


def main():
    steps = int(os.environ.get("SWEEP_STEPS", "3"))
    entries = int(os.environ.get("SWEEP_ENTRIES", str(1 << 20)))
    for spec in sys.argv[1:]:
        label, n_rep, na, nf, dbg = spec.split(":")[:5]
        os.environ["APUS_REP_DBG"] = dbg
        os.environ["SWEEP_PRUNE_BYTES"] = (spec.split(":") + ["0"])[5]
        try:
            d = staged(int(n_rep), entries, 64, 64, steps, int(na), int(nf))
            print(label, n_rep, na, nf, "dbg=" + dbg, json.dumps(brief(d)), flush=True)
        except Exception as exc:      # noqa: BLE001
            print(label, n_rep, na, nf, "dbg=" + dbg, "ERROR", repr(exc)[:300], flush=True)


if __name__ == "__main__":
    main()
