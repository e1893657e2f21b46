#!/usr/bin/env python3
"""HBM traffic of the replica kernels' resident launch, per configuration of the bench line, from two rocprofv3 --pmc passes
each (FETCH_SIZE, WRITE_SIZE; separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of
`python tools/rep_profile_run.py CFG`.
usage: mk_rep_traffic.py CFG FETCH.db WRITE.db FETCH_RUN.log WRITE_RUN.log OUT.json      (adds / replaces CFG in OUT.json)
bytes per entry = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 over the resident launch / the entries that launch committed
(gfx950: FETCH_SIZE counts half of a wide coalesced read, WRITE_SIZE is exact)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mk_traffic import kernel_sum  # noqa: E402
import bench  # noqa: E402


def last_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def main():
    cfg, fdb, wdb, flog, wlog, out = sys.argv[1:7]
    fl, wl = last_line(flog), last_line(wlog)
    assert fl["cfg"] == wl["cfg"] == cfg and fl["verified"] and wl["verified"], (fl, wl)
    assert fl["entries_total"] == wl["entries_total"]
    nf, fetch_kb = kernel_sum(fdb, "FETCH_SIZE", "k_replica")
    nw, write_kb = kernel_sum(wdb, "WRITE_SIZE", "k_replica")
    entries = fl["entries_total"]
    rd, wr = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
    E, N = fl["mean_entry_bytes"], fl["replicas"]
    try:
        doc = json.load(open(out))
    except Exception:
        doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes per configuration of "
                         "`python tools/rep_profile_run.py CFG` (tools/gpu_profile_r5.sh, tools/mk_rep_traffic.py)",
               "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is taken as is",
               "kernel": "k_replica", "configs": {}}
    doc["configs"][cfg] = {
        "replicas": N, "launches": nf, "launches_write_pass": nw, "entries": entries, "mean_entry_bytes": E,
        "FETCH_SIZE_KB_total": fetch_kb, "WRITE_SIZE_KB_total": write_kb,
        "bytes_per_entry": (rd + wr) / entries, "read_bytes_per_entry": rd / entries, "written_bytes_per_entry": wr / entries,
        # what the N rings alone need: every replica's copy written once, the leader's read once for every push
        "ring_bytes_per_entry": (2 * N - 1) * E,
        "launch_ms_under_rocprof": [fl["launch_ms"], wl["launch_ms"]],
        "entries_per_s_under_rocprof": [fl["entries_per_s"], wl["entries_per_s"]]}
    doc["kernel_source_sha256"] = bench.replica_source_hash()     # bench.py quotes these counters only for the build they were taken on
    json.dump(doc, open(out, "w"), indent=1)
    print(cfg, json.dumps({k: round(v, 1) if isinstance(v, float) else v for k, v in doc["configs"][cfg].items()
                           if k in ("bytes_per_entry", "read_bytes_per_entry", "written_bytes_per_entry", "ring_bytes_per_entry", "entries", "launches")}))


if __name__ == "__main__":
    main()
