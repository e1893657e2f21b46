"""Diagnostics for APUS_F_REF_QUIRKS (GPU): the trace tests/traces.py:park_commit_at_wrap one round per call, engine
(default and strict) beside the oracle: offsets / counters at every QUIESCE, and where the per-pass record differs.
  python tools/diag_quirks.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from apus_amd.engine import Engine  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import traces  # noqa: E402
from tests.parity import compare_replica  # noqa: E402


def drive(tr, flags, coalesce):
    cl = orc.Cluster(tr.group_size, tr.log_len, record_apply=True, allow_exact_fit=True)
    eng = Engine(tr.group_size, tr.log_len, flags=flags)
    try:
        eng.reset(); eng.stage_trace(tr)
        reqs = np.ascontiguousarray(tr.reqs, dtype=orc.REQ_DTYPE)
        ev, i, q = tr.events, 0, 0
        while i < len(ev):
            op = ev[i][0]
            if op == "ROUND":
                j = i
                while j < len(ev) and ev[j][0] == "ROUND" and (coalesce or j == i):
                    cl.round(reqs[ev[j][1]:ev[j][1] + ev[j][2]], tr.arena)
                    j += 1
                eng.run_rounds(eng.round_of_g0[ev[i][1]], j - i)
                i = j
                continue
            if op == "PRUNE": cl.tick_prune(); eng.tick_prune()
            elif op == "ELECT": cl.elect(ev[i][1]); eng.elect(ev[i][1])
            elif op == "QUIESCE":
                cl.quiesce(); eng.quiesce()
                held = [r for r in range(eng.group_size) if not (eng.reachable >> r) & 1]
                print(f"  QUIESCE #{q} (event {i}) held={held} status={eng.status_names()}")
                for r in range(tr.group_size):
                    if r in held:
                        continue
                    go, oo = eng.offsets(r), cl.log(r).offsets()
                    line = f"    r{r}: gpu {go} hr={eng.counters(r)['highest_rec']} | orc {oo} hr={cl.highest_rec(r)}"
                    try:
                        compare_replica(eng, cl, r)
                        line += "  EQUAL"
                    except AssertionError as exc:
                        line += "  DIFFERS: " + str(exc).splitlines()[0][:160]
                    print(line)
                q += 1
            elif op == "HOLD": cl.hold(ev[i][1]); eng.hold(ev[i][1])
            elif op == "RELEASE": cl.release(ev[i][1]); eng.release(ev[i][1])
            else: raise ValueError(ev[i])
            i += 1
        gc, ge = eng.round_record()
        oc, oe = cl.round_record()
        bad = np.nonzero((gc != oc) | (ge != oe))[0]
        print(f"  records: {len(gc)} vs {len(oc)}; differ at {bad.tolist()}")
        for b in bad[:16]:
            print(f"    pass {b}: gpu end/commit {ge[b]}/{gc[b]}  orc {oe[b]}/{oc[b]}   (before: gpu {ge[b-1]}/{gc[b-1]} orc {oe[b-1]}/{oc[b-1]})")
    finally:
        eng.close()


if __name__ == "__main__":
    for name in ("park_commit_at_wrap",):
        tr = getattr(traces, name)()
        for flags in (0, 4):
            for coalesce in (False, True):
                print(f"{name} flags={flags} coalesce={coalesce}")
                try:
                    drive(tr, flags, coalesce)
                except Exception as exc:
                    print("  FAILED:", repr(exc)[:300])
    # the flag must not change anything where a quorum exists: the pinned traces in lock step, strict mode
    from apus_amd import trace as T
    from tests.parity import lockstep
    todo = [(n, f()) for n, f in traces.CATALOGUE.items() if n not in ("evict_slow_follower", "no_quorum")]   # (those two reach the 75 % eviction: oracle only)
    todo.append(("steady_small_ring", T.steady_trace(3, 4000, (64, 107), 8, 16, log_len=1 << 15, prune_bytes=1 << 13)))
    for name, tr in todo:
        for coalesce in (True, False):
            cap = max([tr.group_size] + [e[1] + 1 for e in tr.events if e[0] == "JOIN"])
            eng = Engine(tr.group_size, tr.log_len, flags=4, capacity=cap)
            try:
                lockstep(tr, eng, check_at=("QUIESCE",), coalesce=coalesce)
                print(f"lockstep strict {name} coalesce={coalesce}: EQUAL")
            except AssertionError as exc:
                print(f"lockstep strict {name} coalesce={coalesce}: DIFFERS {str(exc).splitlines()[0][:200]}")
            except Exception as exc:
                print(f"lockstep strict {name} coalesce={coalesce}: FAILED {exc!r}"[:300])
            finally:
                eng.close()
