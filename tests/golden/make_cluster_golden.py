"""Writes tests/golden/cluster_ref.json FROM THE REFERENCE ITSELF.

Every trace of tests/traces.py (and BASELINE configs[1] at full size) is replayed on N
instances of the unmodified reference sources (oracle/_ref/libapus_ref_loops.so, built by
`make -C oracle loops` from /root/reference/src/dare/*.c) and the resulting cluster state is
recorded with tests/refparity.cluster_record: per server all 8 offsets, SHA-256 of the defined
ring bytes, SHA-256 of the canonical committed stream, SID, upcall counters, SHA-256 of the
upcall stream; per cluster the leader and the SHA-256 of the per-pass end/commit record.
tests/test_trace_oracle.py::test_cluster_golden_from_reference holds the restated oracle to
these records on any machine (no reference tree needed there).

    python tests/golden/make_cluster_golden.py      # needs /root/reference
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from apus_amd import trace as T            # noqa: E402
from oracle import refloops                # noqa: E402
from tests import traces                   # noqa: E402
from tests.refparity import cluster_record  # noqa: E402


def main():
    if not refloops.available():
        raise SystemExit("oracle/_ref/libapus_ref_loops.so cannot be built here (no /root/reference)")
    out = {"generator": "tests/golden/make_cluster_golden.py",
           "source": "reference-as-is: /root/reference/src/dare/*.c compiled unmodified (oracle/Makefile `loops`)",
           "cases": {}}
    cases = dict(traces.CATALOGUE)
    cases["c2_full"] = T.config_c2
    cases["c5_rejoin_full"] = lambda: T.config_c5(rejoin=True)      # BASELINE configs[4] at full size incl. the JOIN tail
    for name in sorted(cases):
        tr = cases[name]()
        rc = refloops.run_trace(tr)
        out["cases"][name] = {"group_size": tr.group_size, "log_len": tr.log_len, "n_reqs": int(tr.n_reqs),
                              "n_events": len(tr.events), "record": cluster_record(rc, rc.n)}
        rc.close()
        print(name, "leader", out["cases"][name]["record"]["leader"], "passes", out["cases"][name]["record"]["rounds"])
    with open(os.path.join(HERE, "cluster_ref.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
