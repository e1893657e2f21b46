#!/usr/bin/env python3
"""HBM traffic of k_replica from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) of `python tools/rep_bench.py --grid A:F --steps K --no-hostfed --brief`.
usage: mk_rep_traffic.py FETCH.db WRITE.db RUN.log OUT.json
bytes per entry = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 over the resident launch / the entries that launch committed
(gfx950: FETCH_SIZE counts half of a wide coalesced read, WRITE_SIZE is exact)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mk_traffic import kernel_sum  # noqa: E402
import bench  # noqa: E402


def main():
    fdb, wdb, log, out = sys.argv[1:5]
    line = json.loads([l for l in open(log) if l.startswith("{")][-1])
    nf, fetch_kb = kernel_sum(fdb, "FETCH_SIZE", "k_replica")
    nw, write_kb = kernel_sum(wdb, "WRITE_SIZE", "k_replica")
    entries = line["entries_total"]
    total = (2.0 * fetch_kb + write_kb) * 1024.0
    doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes of "
                     "`python tools/rep_bench.py --grid A:F --steps K --no-hostfed --brief` (tools/gpu_profile.sh REPLICA=1, tools/mk_rep_traffic.py)",
           "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is taken as is",
           "kernel": "k_replica", "replicas": line["replicas"], "grid": [line["n_append"], line["n_fwork"]],
           "launches": nf, "FETCH_SIZE_KB_total": fetch_kb, "WRITE_SIZE_KB_total": write_kb,
           "entries": entries, "bytes_total": total, "bytes_per_entry": total / entries,
           "read_bytes_per_entry": 2.0 * fetch_kb * 1024.0 / entries, "written_bytes_per_entry": write_kb * 1024.0 / entries,
           "entries_per_s_under_rocprof": line["entries_per_s"],
           "kernel_source_sha256": bench.replica_source_hash()}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("bytes_per_entry", "read_bytes_per_entry", "written_bytes_per_entry", "entries", "launches")}))


if __name__ == "__main__":
    main()
