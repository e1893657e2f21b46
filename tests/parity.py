"""Lock-step comparison of the GPU engine with the CPU oracle on one trace.

The engine is driven through the C ABI (apus_amd.engine.Engine -> libapus_gpu.so);
the oracle (oracle/liboracle.so) is only the checker."""
from __future__ import annotations

import numpy as np

from oracle import oracle as orc


def compare_replica(eng, cl, r, tag=""):
    go, oo = eng.offsets(r), cl.log(r).offsets()
    assert go == oo, f"{tag} replica {r}: offsets differ\n gpu={go}\n orc={oo}"
    ring_o = cl.log(r).ring()
    end, head = oo["end"], oo["head"]
    if end != oo["len"]:
        ring_g = eng.ring(r)
        mask = orc.defined_mask(ring_o, end, head, end)
        diff = np.nonzero(ring_g[mask] != ring_o[mask])[0]
        if len(diff):
            pos = np.nonzero(mask)[0][diff[:8]]
            raise AssertionError(f"{tag} replica {r}: {len(diff)} defined ring bytes differ, first at {pos.tolist()} "
                                 f"gpu={ring_g[pos].tolist()} orc={ring_o[pos].tolist()} (head={head}, end={end})")
        hg = orc.canon_hash(ring_g, end, head, oo["commit"])
        ho = orc.canon_hash(ring_o, end, head, oo["commit"])
        assert hg == ho, f"{tag} replica {r}: canonical digest differs {hg} vs {ho}"
    w = eng.hdr_words(r)
    c = eng.counters(r)
    assert c["highest_rec"] == cl.highest_rec(r), f"{tag} replica {r}: highest_rec {c['highest_rec']} vs {cl.highest_rec(r)}"
    assert int(w[16]) == cl.apply_count(r), f"{tag} replica {r}: apply_count {int(w[16])} vs {cl.apply_count(r)}"
    assert c["apply_hash"] == cl.apply_hash(r), f"{tag} replica {r}: apply stream hash differs"
    assert int(w[21]) == cl.store_count(r), f"{tag} replica {r}: store_count {int(w[21])} vs {cl.store_count(r)}"
    assert int(w[17]) == cl.log(r).prev_head, f"{tag} replica {r}: prev_head"
    assert c["sid"] == cl.sid(r), f"{tag} replica {r}: sid {c['sid']:#x} vs {cl.sid(r):#x}"


def compare_all(eng, cl, tag="", replicas=None):
    for r in (range(eng.group_size) if replicas is None else replicas):
        compare_replica(eng, cl, r, tag)
    gc, ge = eng.round_record()
    oc, oe = cl.round_record()
    assert len(gc) == len(oc), f"{tag}: round count {len(gc)} vs {len(oc)}"
    bad = np.nonzero((gc != oc) | (ge != oe))[0]
    assert len(bad) == 0, (f"{tag}: per-round record differs at rounds {bad[:8].tolist()}: "
                           f"gpu end/commit={ge[bad[:4]].tolist()}/{gc[bad[:4]].tolist()} "
                           f"orc={oe[bad[:4]].tolist()}/{oc[bad[:4]].tolist()}")


def compare_apply_tail(eng, cl, r, last=512):
    """Full record-by-record comparison of the newest `last` apply upcalls."""
    oa = cl.apply_log(r)
    if len(oa) == 0:
        return
    oa = oa[-last:]
    lo, hi = int(oa["slot"][0]), int(oa["slot"][-1])
    ga = eng.apply_records(r, lo, hi - lo + 1)
    ga = ga[ga["kind"] != 0]
    assert len(ga) == len(oa), f"replica {r}: {len(ga)} apply records vs {len(oa)}"
    for name in oa.dtype.names:
        assert np.array_equal(ga[name], oa[name]), f"replica {r}: apply field {name} differs"


def lockstep(trace, eng, check_at=("PRUNE", "QUIESCE"), coalesce=True, allow_exact_fit=True, batch=False):
    """Feed the same events to the engine and to a fresh oracle cluster; compare
    at every quiescent event.  Returns the oracle cluster.
    batch=True: stretches of ROUND / PRUNE events go through apus_gpu_batch_begin/_end
    (multi-segment launches); use check_at=("QUIESCE",) so that stretches span prune ticks."""
    cl = orc.Cluster(trace.group_size, trace.log_len, record_apply=True, allow_exact_fit=allow_exact_fit)
    eng.reset()
    eng.stage_trace(trace)
    reqs = np.ascontiguousarray(trace.reqs, dtype=orc.REQ_DTYPE)
    ev = trace.events
    i = 0
    opened = False

    def batch_open():
        nonlocal opened
        if batch and not opened:
            eng.batch_begin(); opened = True

    def batch_close():
        nonlocal opened
        if opened:
            eng.batch_end(); opened = False

    while i < len(ev):
        op = ev[i][0]
        if op == "ROUND":
            j = i
            while j < len(ev) and ev[j][0] == "ROUND" and (coalesce or j == i):
                cl.round(reqs[ev[j][1]:ev[j][1] + ev[j][2]], trace.arena)
                j += 1
            batch_open()
            eng.run_rounds(eng.round_of_g0[ev[i][1]], j - i)
            i = j
            continue
        if op == "PRUNE" and op not in check_at:
            batch_open()
        else:
            batch_close()
        if op == "ELECT":
            cl.elect(ev[i][1]); eng.elect(ev[i][1])
        elif op == "PRUNE":
            cl.tick_prune(); eng.tick_prune()
        elif op == "QUIESCE":
            cl.quiesce(); eng.quiesce()
        elif op == "HOLD":
            cl.hold(ev[i][1]); eng.hold(ev[i][1])
        elif op == "RELEASE":
            cl.release(ev[i][1]); eng.release(ev[i][1])
        elif op == "KILL":
            cl.kill(ev[i][1]); eng.kill(ev[i][1])
        elif op == "JOIN":
            cl.join(ev[i][1]); eng.join(ev[i][1])
            assert cl.n == eng.group_size
        else:
            raise ValueError(ev[i])
        if op in check_at:
            if op == "PRUNE":
                # the oracle's followers learn the newest commit lazily (one poll
                # later, dare_ibv_rc.c:1761-1819); settle both sides before comparing
                cl.quiesce(); eng.quiesce()
            eng.check_status()
            held = [r for r in range(eng.group_size) if not (eng.reachable >> r) & 1]
            compare_all(eng, cl, tag=f"event {i} {ev[i]}",
                        replicas=[r for r in range(eng.group_size) if r not in held])
        i += 1
    batch_close()
    eng.check_status()
    assert cl.force_prunes == 0, ("the trace fills the log to 75 %: the reference's force_log_pruning "
                                  "(dare_server.c:2069, follower eviction) is modelled by the oracle only")
    return cl


def step_commands(trace, round_of_g0):
    """the commands of ONE step of a steady trace, as bench.py issues them to the replica kernels: a stretch of consecutive
    ROUND events = ("run", first staged round, rounds), a PRUNE event = ("prune",)"""
    out, ev, i = [], trace.events, 0
    while i < len(ev):
        if ev[i][0] == "ROUND":
            j = i
            while j < len(ev) and ev[j][0] == "ROUND":
                j += 1
            out.append(("run", round_of_g0[ev[i][1]], j - i))
            i = j
            continue
        if ev[i][0] == "PRUNE":
            out.append(("prune",))
        i += 1
    return out


def oracle_replay_steps(trace, steps, record_last=True):
    """The oracle's side of a bench-shaped run (bench.py: measure_replica_kernels, dare_server.c:1012-1125 is the loop it
    stands for): ELECT, then `steps` passes over the trace's ROUND / PRUNE events on the same logs, then QUIESCE.  The apply
    upcalls are recorded for the last step only (a hundred steps of configs[1] are 10^8 upcalls).  -> the cluster"""
    rounds = [(e[1], e[2]) for e in trace.events if e[0] == "ROUND"]
    round_of_g0 = {g0: i for i, (g0, _) in enumerate(rounds)}
    round_n = np.array([n for _, n in rounds], dtype=np.uint32)
    first = np.concatenate([[0], np.cumsum(round_n)]).astype(np.int64)
    reqs = np.ascontiguousarray(trace.reqs, dtype=orc.REQ_DTYPE)
    arena = np.ascontiguousarray(trace.arena, dtype=np.uint8)
    cmds = step_commands(trace, round_of_g0)
    cl = orc.Cluster(trace.group_size, trace.log_len, record_apply=False)
    cl.elect(trace.leader)
    for s in range(steps):
        if record_last and s == steps - 1:
            cl.L.orc_cluster_record_apply(cl.h, 1)
        for c in cmds:
            if c[0] == "run":
                r0, n = c[1], c[2]
                cl.run_rounds(reqs[first[r0]:first[r0 + n]], round_n[r0:r0 + n], arena, 0)
            else:
                cl.tick_prune()
    cl.quiesce()
    return cl


def lockstep_rep(trace, eng, source="staged", **kw):
    """The trace through the replica kernels (Engine.run_trace_rep) with the oracle in lock step: behind every QUIESCE
    event -- the run is parked there -- every reachable replica is compared bit for bit.  Returns the oracle cluster."""
    cl = orc.Cluster(trace.group_size, trace.log_len, record_apply=True)
    reqs = np.ascontiguousarray(trace.reqs, dtype=orc.REQ_DTYPE)
    arena = np.ascontiguousarray(trace.arena, dtype=np.uint8)
    ev = trace.events
    done = [0]                      # oracle: events carried out

    def catch_up(upto):
        while done[0] <= upto:
            e = ev[done[0]]
            op = e[0]
            if op == "ROUND": cl.round(reqs[e[1]:e[1] + e[2]], arena)
            elif op == "ELECT": cl.elect(e[1])
            elif op == "PRUNE": cl.tick_prune()
            elif op == "QUIESCE": cl.quiesce()
            elif op == "KILL": cl.kill(e[1])
            elif op == "HOLD": cl.hold(e[1])
            elif op == "RELEASE": cl.release(e[1])
            elif op == "JOIN": cl.join(e[1])
            else: raise ValueError(e)
            done[0] += 1

    def on_event(i, e, engine):
        catch_up(i)
        if e[0] == "QUIESCE":
            engine.check_status()
            for r in range(engine.group_size):
                if (engine.reachable >> r) & 1:
                    compare_replica(engine, cl, r, tag=f"replica kernels ({source}), event {i} {e}")
    eng.run_trace_rep(trace, source=source, on_event=on_event, **kw)
    catch_up(len(ev) - 1)
    assert cl.force_prunes == 0
    return cl
