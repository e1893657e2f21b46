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
